/* kxp_format.h — layout of a compiled Kleenex program blob ("KXP").
 *
 * The blob is the contract between the compiler back end and the execution
 * engines; it takes the place of the reference's IL `Program`
 * (src/KMC/Program/IL.hs:69-90) once printed to tables:
 *
 *   register form — what IL blocks mean for `--la=false` programs: per state a
 *     byte-class → (next state, action) map, actions being ordered micro-ops
 *     ResetI / AppendI / AppendSymI / ConcatI (IL.hs:40-47) over registers,
 *     register 0 being the stream buffer (progStreamBuffer);
 *   path form — the same transducer with the registers resolved away: per
 *     transition and per leaf of the target path tree, the source leaf it
 *     extends and the bytes appended.  This is what the HIP
 *     engine executes; both forms describe the same function.
 *
 * All integers little-endian; every array is padded to a multiple of 4 bytes.
 *
 *   char  magic[8] = "KXPBLOB1"
 *   u32   version (=1), nstages, info_len ; u8 info[info_len]
 *   per stage:
 *     u32 'KXST', nstates, nclasses, q0, nregs, nactions, nops, nconsts, constpool_len,
 *         maxleaves, nback, npconsts, pconstpool_len, nsync, sync_complete, actions
 *     u8  cls[256]
 *     u16 delta[nstates*nclasses]            0xFFFF = no transition (FailI)
 *     u32 act[nstates*nclasses]              action id
 *     u32 final_act[nstates]                 0xFFFFFFFF = not final
 *     u32 act_off[nactions+1]
 *     {u32 op<<24|dst ; u32 arg} ops[nops]
 *     u32 const_off[nconsts+1] ; u8 constpool[]
 *     u32 pback[nstates*nclasses]            backward-row id of the transition
 *     u8  nleaves[nstates] ; u8 fin_leaf[nstates]   (0xFF = not final)
 *     u32 back[nback*maxleaves]              parent | copy<<8 | pconst<<9 ; 0xFFFFFFFF dead
 *                                            (stages with symbol tables: pconst < 2^15, bits 24-31 = table+1, 0 = none)
 *     u32 pconst_off[npconsts+1] ; u8 pconstpool[]
 *     u32 init_const[maxleaves]              pconst per leaf of q0's closure
 *     u32 sync_next[nsync*nclasses] ; u32 sync_state[nsync]
 *     if (actions & KXP_STAGE_HAS_TABLES): u32 ntables ; u8 table[ntables][256]
 *
 * Symbol tables (round 3) — the IL's AppendTblI (IL.hs:44, SSTCompiler/Classes.hs:102-125, C.hs:421-430): output =
 * table[symbol], one byte.  Register form: micro-op KXP_OP_APPEND_TBL dst, table.  Path form: a back entry whose copy bit
 * is set and whose table field is t+1 appends table[t][input byte] instead of the input byte.
 *
 * Register actions (`r@t`, `!r`, `[r <- …]`, `[r += …]`; Kleenex/Actions.hs:28-38).  `actions` bit 0 clear: the stage's output is
 * final.  `actions` = 1 | nregs << 8 (| KXP_STAGE_HAS_TABLES): the stage's output is a TOKEN STREAM that an action interpreter replays on a stack of
 * buffers and nregs registers before it leaves the stage (action post-pass) — escape byte 0xFF:
 *     FF FF   the byte 0xFF            FF 00   Push: a new empty buffer on the stack
 *     FF 01 r Pop r: register r := top buffer, popped            FF 02 r Write r: top buffer ++= register r; r := empty
 *     any other byte: appended to the top buffer.   The stage's output is the bottom buffer.
 */
#ifndef KXP_FORMAT_H
#define KXP_FORMAT_H

#define KXP_MAGIC "KXPBLOB1"
#define KXP_VERSION 1u
#define KXP_STAGE_MAGIC 0x5453584Bu /* 'KXST' */

#define KXP_NO_STATE 0xFFFFu
#define KXP_NOT_FINAL 0xFFFFFFFFu
#define KXP_DEAD_LEAF 0xFFFFFFFFu
#define KXP_NO_LEAF 0xFFu

/* micro-ops of the register form (IL.hs:40-47) */
#define KXP_OP_RESET 0u        /* ResetI  dst           */
#define KXP_OP_APPEND_CONST 1u /* AppendI dst, const    */
#define KXP_OP_APPEND_SYM 2u   /* AppendSymI dst, 0     */
#define KXP_OP_CONCAT 3u       /* ConcatI dst, src      */
#define KXP_OP_APPEND_TBL 4u   /* AppendTblI dst, table: appends table[symbol] */

#define KXP_STAGE_HAS_TABLES 2u /* bit of the stage header's `actions` word: a symbol-table section follows sync_state */
#define KXP_MAX_TABLES 254u
#define KXP_ENGINE_TABLES 7u /* tables one stage may use beside each other on the engine (3-bit table field of a path entry);
                                a compiler with more writes the smaller ones out as constants */

/* sync_state values for non-singleton subsets */
#define KXP_SYNC_MULTI 0xFFFFFFFFu   /* several states still possible */
#define KXP_SYNC_EMPTY 0xFFFFFFFEu   /* every start state has failed */
#define KXP_SYNC_UNKNOWN 0xFFFFFFFDu /* subset construction was capped here */

/* action tokens (second byte after KXP_ESC) */
#define KXP_MAX_ACTION_REGS 251u /* registers of a stage's action interpreter: 0..250 (a register index is one token byte) */
#define KXP_ESC 0xFFu
#define KXP_TOK_PUSH 0x00u
#define KXP_TOK_POP 0x01u
#define KXP_TOK_WRITE 0x02u

#endif
