// kernels_acl.hpp -- arithmetic coder, one 10 MiB block per LANE (ac_encode_lanes_k).
// Reference: ac_coder::{write,flush} (/root/reference/arithmetic.cpp:108-169), block framing of ac_write (:318-363).
//
// ac_encode_k / ac_encode_rows_k (kernels_ac.hpp) spend a wavefront on 1, 4 or 8 blocks: lowest latency of a block, but
// 56..63 of the 64 lanes execute garbage in every step, and a launch of three 50 M-read shards occupies 179 CUs at 59 %
// for as long as it runs.  Here every lane of the chain wave carries its OWN block: the same ~13 instructions of a step
// serve 64 blocks, nothing moves between lanes (no DPP, no hand-over), and 1431 blocks are 23 workgroups of three waves --
// the chip is left to the front stages of the other shards in flight.  A block's latency stays what it was (one step is
// still ~16 issue slots of one wavefront).
//
// What makes a lane self-sufficient is the form of the output.  The reference writes bits one at a time with a count of
// pending "underflow" bits (arithmetic.cpp:133-147); that count is carry propagation in disguise: the coded block is the
// binary expansion of
//     X = raw16 . sum_i  B_i * 2^-(16 + S_i + 32),     S_i = t_0 + .. + t_(i-1),
// B_i = floor(range_i * c_lo / total) the offset step i adds to `lo`, t_i the bits step i renormalises by -- closed by
// the flush of :160-169, which keeps S + 2 bits behind the raw symbols: (X cut there | 1) + bit 30 of the final lo.
// (Checked against the oracle's coder, incl. heavy-underflow tables, before this kernel was written; the parity tests
// drive carries through words that were already stored.)  So the bit sink of a block is a 96-bit accumulator in three
// registers: add B at the current bit offset, move on by t, store a word whenever 32 bits are complete; a carry beyond
// the last complete word (it needs 32 ones in a row there) walks back through the stored words, rarely.
//
// Workgroup = four wavefronts (one per SIMD of a CU) over the same 64 blocks, rounds of 16 symbols per block:
//   gather  lane b reads block b's symbols 16 at a time, forms the contexts and fetches the table rows
//           {g(c_lo), g(c_hi)} (kernels_ac.hpp: reciprocal fractions) three rounds ahead, straight into LDS
//           (global_load_lds_dwordx4: the row of lane b lands at ops[slot][step][b], no register in between)
//   chain   16 x { ds_read_b128 operands, 14 VALU: A, B, D, nlo, renorm count, new (lo, M), Q += t ; ds_write_b64 (B, Q) }
//           Q = 16 + bits dropped so far = where the top bit of the next B belongs in the block's bit stream.
//           Rounds that are not plain for a lane -- the first (two raw symbols), a block's tail, a symbol that is the
//           last of its context, a range that renormalises to the full 2^32 -- are redone for those lanes by the
//           general step under the exec mask
//   sink    lane b adds block b's (B, Q) records of the previous round into its accumulator and reports, per step, the word
//           that may still take a carry and its index (two steps per 16-byte record: round 4 -- in round 3 this wave stored
//           the word into the staging ring itself, an LDS instruction and two of address arithmetic per step of the wave
//           that sets the pace of a round)
//   writer  puts the reported words into its staging ring, in step order, and moves those that are final from the ring to
//           the block's output, a 128-byte line per lane and visit
// One LDS-only barrier per round.  Contexts depend on symbols only, never on coder state, which is why the gather
// wave can run ahead.  Measured costs behind this split: tools/ubench_lds.hip (a DS instruction costs a wave 17-35 cycles
// whatever its width, a branch on a VALU result ~30, a VALU instruction 4), DESIGN.md section 5.
// Only for tables whose largest context total is <= 2^29 (the host checks, as for the plain path of ac_encode_k):
// every symbol then keeps an interval of at least two values, and the reference's coder never runs into the inverted
// intervals whose bits are not those of X.
#pragma once
#include "kernels_ac.hpp"

namespace scalce {

constexpr int ACL_STEPS = 16;  // symbols per block and round

// general step on a well-formed state, reporting what the sink needs: B = offset added to lo, t = bits dropped
__device__ __forceinline__ void acl_step_general(u32 &lo, u32 &hi, const uint4 g, u32 &B, u32 &t) {
  const u32 R = hi - lo;
  u32 M;
  const bool wrap = __builtin_add_overflow(R, 1u, &M);
  const u32 qa = mulfrac_m(M, wrap, g.z, g.w);
  const u32 qb = mulfrac_m(M, wrap, g.x, g.y);
  const u32 nhi = (g.w == 0xFFFFFFFFu) ? hi : lo + qa - 1;  // c_hi == total: hi unchanged
  const u32 nlo = lo + qb;
  B = qb;
  const u32 x = nlo ^ nhi;
  const u32 kf = x ? (u32)__clz(x) : 31u;  // x == 0 needs a context total above 2^29: excluded by the host
  const u32 ks = kf < 31u ? kf : 31u;
  const u32 z = ((nlo & ~nhi) << ks) << 1;
  const u32 u = (u32)__clz(~z);
  t = ks + u;
  lo = (nlo << t) & 0x7FFFFFFFu;
  hi = (nhi << t) | ((1u << t) - 1) | 0x80000000u;
}

// plain step on (lo, M), M = hi - lo + 1 modulo 2^32 (sys_step of kernels_ac.hpp without the travelling): lo keeps a
// stray bit 31 (it changes neither B nor t), M = 0 is absorbing (D + 1 = 0 whatever the operands) and is what the
// round's exit test looks for afterwards
__device__ __forceinline__ void acl_step_plain(u32 &lo, u32 &M, const uint4 g, u32 ones, u32 zero, u32 &B, u32 &t) {
  const u32 A1 = (u32)(((u64)M * g.w + (((u64)ones << 32) | __umulhi(M, g.z))) >> 32);
  B = (u32)(((u64)M * g.y + (((u64)zero << 32) | __umulhi(M, g.x))) >> 32);
  const u32 D = A1 - B;
  const u32 nlo = lo + B;
  t = renorm_count(nlo, D);
  M = (D + 1) << t;
  lo = nlo << t;
}

// Bit sink of one block in one lane.  [w2 w1 w0] is a 96-bit window of X: with Qb the stream position of the next B's
// top bit, w1 is the block's word Qb >> 5, w2 the word in front of it -- complete, kept back because a carry may still
// reach it -- and B's top bit sits Qb & 31 bits below the top of w1.
//
// What an instruction costs a lone wavefront here (tools/ubench_lds.hip): a VALU instruction 4 cycles; a DS instruction
// 17 (read) to 25-35 (write) whatever its width; a scattered global store ~100 with all lanes, ~40 with four; and every
// trip from a VALU result through an SGPR into a scalar instruction or a branch -- s_and_saveexec, s_cbranch_vccz -- ~16
// on top.  As first written, with the natural ifs (carry? word complete? room?), a step was 55 instructions, seven
// branches and a masked store: 330 cycles, and the chain wave waited for the sink 66 % of its time.  Hence the shape of
// step(): fourteen VALU instructions and ONE LDS store, no branch, no exec mask --
//   * w2 goes to the staging ring in every step, complete or not (a later step overwrites it); the writer wave takes the
//     final words of all lanes out, 16 bytes per lane and store;
//   * a carry out of w2 (it needs w2 = 0xFFFFFFFF) is only COUNTED; a lane that counted one redoes its round from the
//     saved state with careful(), which notes where the carry belongs;
//   * the notes -- word indices, in a log that grows down from the end of the block's own output buffer -- are applied by
//     finish(): additions commute.  (Resolved on the spot, a loop of loads and stores in the step sequence, the compiler
//     put s_waitcnt vmcnt(0) in front of every store of the following steps.)
// staging ring of the writer wave, words per lane: it takes whole 128-byte lines (32 words) out and puts at most 16 words
// per round in front of them: 31 + 16 + 16 + 2 <= WORDS.
template <int LW, int SETS> struct AclRing { static constexpr int WORDS = 64; };
template <int LW>
struct AclSink {
  SCALCE_GLOBAL u32 *dst;
  u32 wcap;      // words the block may write
  SCALCE_GLOBAL u32 *logtop;  // the carry notes grow DOWN from here: the last word of the block's own buffer, or of a log of its own
  u32 logcap;    // ... and how many there may be (0: they share the block's buffer with the coded words)
  u32 w2, w1, w0;
  u32 Qb;        // stream position of the next B's top bit
  u32 ncar;      // smallest w2 seen right after a step of this round added its carry: 0 = a carry may have left w2 (step())
  u32 nlog;      // notes in the log
  bool over;
  __device__ __forceinline__ void init(SCALCE_GLOBAL u32 *d, u32 cap_words, SCALCE_GLOBAL u32 *log, u32 log_words, u32 s0, u32 s1) {
    dst = d; wcap = cap_words;
    logtop = log ? log + (log_words - 1u) : d + (cap_words - 1u);
    logcap = log ? log_words : 0u;
    w2 = 0; w1 = (s0 << 24) | (s1 << 16); w0 = 0;  // the two raw symbols (arithmetic.cpp:110-120): 16 bits of X
    Qb = 16; ncar = ~0u; nlog = 0; over = false;    // (w2 = the empty word in front of the block: ring slot 0, never taken out)
  }
  __device__ __forceinline__ u32 final_words() const {  // the words in front of w2 are final
    const u32 wq = Qb >> 5;
    return wq ? wq - 1u : 0u;
  }
  // room for a round's words and notes?  (once per round: a lane that runs out of room stops and reports)
  __device__ __forceinline__ bool room() {
    if ((Qb >> 5) + 2 * ACL_STEPS + 2 + (logcap ? 0u : nlog) >= wcap) over = true;
    if (logcap && nlog + ACL_STEPS + 2 >= logcap) over = true;
    return !over;
  }
  // the step: X += B at bit Qb, then on to Qa.  Reports the word that may still take a carry as it stands now (val) and
  // the index wq of the word behind it: word wq - 1 of the block is `val` until a later step says otherwise -- the writer
  // wave puts it into the staging ring (slot wq & (ring words - 1)); it is final once a step completes word wq.
  __device__ __forceinline__ void step(u32 B, u32 Qa, u32 &val, u32 &wq) {
    const u32 pos = Qb & 31u;
    const u32 b_hi = B >> pos;
    const u32 b_lo = __builtin_amdgcn_alignbit(B, 0u, pos);  // low word of {B, 0} >> pos: B << (32 - pos), 0 for pos = 0
    u32 c0, c1;
    w0 = __builtin_addc(w0, b_lo, 0u, &c0);
    w1 = __builtin_addc(w1, b_hi, c0, &c1);
    // A carry out of w2 leaves w2 = 0 behind.  Instead of a fourth link in the carry chain (and the wait state in front of
    // it) the step keeps the smallest w2 it has seen: a round that saw 0 is redone by careful().  (A word that is 0 for
    // another reason -- the empty word in front of the block, a genuine all-zero word -- only costs that redo.)
    w2 += c1;
    ncar = w2 < ncar ? w2 : ncar;
    wq = Qb >> 5;
    val = w2;
    const bool f = (Qa >> 5) != wq;
    w2 = f ? w1 : w2;
    w1 = f ? w0 : w1;
    w0 = f ? 0u : w0;
    Qb = Qa;
  }
  // the same step for a round in which step() counted a carry out of w2: it belongs to the word in front of w2
  __device__ __forceinline__ void careful(u32 B, u32 Qa) {
    const u32 pos = Qb & 31u;
    const u32 b_hi = B >> pos;
    const u32 b_lo = __builtin_amdgcn_alignbit(B, 0u, pos);
    u32 c0, c1, c2;
    w0 = __builtin_addc(w0, b_lo, 0u, &c0);
    w1 = __builtin_addc(w1, b_hi, c0, &c1);
    w2 = __builtin_addc(w2, 0u, c1, &c2);
    const u32 wq = Qb >> 5;
    if (c2 != 0u) {
      *(logtop - nlog) = wq - 2u;  // (wq >= 2: a carry out of w2 needs 32 ones there, the empty word in front of the block has none)
      nlog++;
    }
    if ((Qa >> 5) != wq) { w2 = w1; w1 = w0; w0 = 0u; }  // (the words are those step() reported: only the note is new)
    Qb = Qa;
  }
  static __device__ __forceinline__ void carry_back(SCALCE_GLOBAL u32 *dst, u32 wcap, int k) {
    for (; k >= 0; k--) {
      if ((u32)k >= wcap) continue;
      const u32 v = __builtin_bswap32(__hip_atomic_load(&dst[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
      __hip_atomic_store(&dst[k], __builtin_bswap32(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v) break;
    }
  }
  // flush (arithmetic.cpp:160-169) in terms of X: cut behind the bit after B's top position, set that bit, add bit 30 of
  // the final lo there.  The words below w2 are in global memory by now (writer wave).  Returns the block's size in bytes.
  __device__ __forceinline__ u32 finish(u32 final_lo) {
    const u32 pos = Qb & 31u;
    const int wi = (int)(Qb >> 5) - 1;  // index of w2's word, -1 for a block of a few symbols
    const u64 ulp = 1ull << (62u - pos);
    u64 v = ((u64)w1 << 32) | w0;
    v = (v & ~(ulp - 1)) | ulp;
    if ((final_lo >> 30) & 1u) {
      const bool c = __builtin_add_overflow(v, ulp, &v);
      if (c) {
        w2 += 1u;
        if (w2 == 0u && !over) { *(logtop - nlog) = (u32)(wi - 1); nlog++; }
      }
    }
    const u32 bits = Qb + 2u;
    const u32 words[3] = {w2, (u32)(v >> 32), (u32)v};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int idx = wi + k;
      if (idx >= 0 && 32u * (u32)idx < bits) {
        if ((u32)idx < wcap) dst[idx] = __builtin_bswap32(words[k]);
        else over = true;
      }
    }
    if (nlog && !over) {  // the carries noted on the way, into the words as they stand now
      __threadfence();
      for (u32 e = 0; e < nlog; e++) {
        const int k = (int)__hip_atomic_load(logtop - e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        carry_back(dst, wcap, k);
      }
    }
    return (bits + 7) >> 3;
  }
};

// LW = lanes (blocks) of a set of four waves, SLOTS = operand ring: the gather wave runs SLOTS - 1 rounds ahead of the chain
template <int LW, int SLOTS, int RINGW>
struct AclShared {
  uint4 ops[SLOTS][ACL_STEPS][LW];      // gather -> chain: operands of a round (slot = round % SLOTS), written by LDS-direct loads
  uint4 rec[2][ACL_STEPS / 2][LW];      // chain -> sink: (Q, B) of two steps of a round per entry
  uint4 fifo[2][ACL_STEPS / 2][LW];     // sink -> writer: (val, wq) of two steps per entry (AclSink::step)
  u32 stage[RINGW][LW];                 // writer: coded words on their way out -- word k of the block in slot (k + 1) & (WORDS - 1)
  u32 pub[2][LW];                       // sink -> writer: words below this index are final
  u32 final_lo[LW];
};

// VMEM of the gather wave is issued and awaited by hand.  Its loads are consumed two to four rounds after they were
// requested, across the loop's back edge, where the compiler's own s_waitcnt placement falls back to vmcnt(0) -- a full
// memory round trip exposed in every other round (the first version of this kernel: 110 cycles per step in the gather
// wave).  The compiler never sees these loads, so it inserts nothing; the counts below are exact because the wave issues
// no other vector memory instruction.
__device__ __forceinline__ u32x4 acl_load16(const SCALCE_GLOBAL u8 *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void acl_wait_vm(u32x4 &v) {  // ... until at most N younger VMEM instructions are outstanding
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void acl_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// SETS = 1: one set of four waves per workgroup, a wave per SIMD (round 3).  SETS = 2 (round 4): TWO sets in a workgroup of
// eight waves over 2 x LW blocks, two waves per SIMD -- a chain or sink wave spends half of a step waiting for its DS
// instructions, slots that a light wave (gather, writer) of the OTHER set takes: wave w of a workgroup
// goes to SIMD w % 4, so
//     SIMD 0: chain A + gather B    SIMD 1: chain B + gather A    SIMD 2: sink A + writer B    SIMD 3: sink B + writer A
// A CU then codes 2 x LW blocks at (nearly) the pace of LW.  Each wave claims 256 registers: two per SIMD, nothing else.
// LDS holds both sets (LW = 48, SLOTS = 3: 2 x 73 KB).  One barrier per round for all eight waves.
template <bool EXCLUSIVE, int SETS, int LW, int SLOTS>
__global__ __launch_bounds__(256 * SETS) void ac_encode_lanes_k(AcEncArgs a) {
  static_assert(SLOTS >= 3 && SLOTS <= 9 && LW <= 64 && (SETS == 1 || SETS == 2), "");
  __shared__ AclShared<LW, SLOTS, AclRing<LW, SETS>::WORDS> shs[SETS];
  const int lane = lane_id();
  const int w = wave_id();
  // SETS == 1: 0 chain, 1 gather, 2 sink, 3 writer: the four SIMDs of the CU.  SETS == 2: see above.
  // (a.pairing, experiments: 1 = chain + writer / sink + gather, 2 = chain + sink / gather + writer on a SIMD)
  const u32 rtab = a.pairing == 1 ? 0x11332200u : a.pairing == 2 ? 0x33221100u : 0x33112200u;  // role of wave w: nibble w
  const u32 stab = a.pairing == 1 ? 0x01011010u : a.pairing == 2 ? 0x01011010u : 0x01011010u;  // set of wave w
  const int role = SETS == 1 ? w : (int)((rtab >> (4 * w)) & 15u);
  const int set = SETS == 1 ? 0 : (int)((stab >> (4 * w)) & 15u);
  AclShared<LW, SLOTS, AclRing<LW, SETS>::WORDS> &sh = shs[set];
  // Workgroups are dealt to the 8 XCDs in turn (b and b + 8 share one: MI355X_MICROARCH.md, Workgroup dispatch), and
  // every XCD has an L2 of its own.  A launch holds the blocks of several streams back to back, each stream with its own
  // 4 MB table: the workgroups of an XCD take CONSECUTIVE groups of blocks, so that an L2 serves one or two tables
  // instead of all of them (with three tables per L2 the gather wave, not the chain, set the pace of a launch).  Speed
  // only: any placement gives the same bytes.
  const u32 nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7u, xcd = blockIdx.x & 7u;
  const u32 wg = (xcd < xr ? xcd * (xq + 1u) : xr * (xq + 1u) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const u32 bpw = a.lanes_used && a.lanes_used < (u32)LW ? a.lanes_used : (u32)LW;  // blocks per set (lanes in use)
  const u32 blk = (wg * SETS + (u32)set) * bpw + (u32)lane;
  const bool have = blk < a.nblocks && (u32)lane < bpw;
  const SCALCE_GLOBAL AcBlockDesc *dp = (const SCALCE_GLOBAL AcBlockDesc *)a.desc + (have ? blk : 0u);
  const u32 n = have ? dp->n : 0u;
  const u32 nr = (n + ACL_STEPS - 1) / ACL_STEPS;  // rounds of this lane's block
  u32 nr_wg = nr;                                    // rounds of the workgroup = those of its longest block
  if (SETS == 2) {                                   // (the other set's blocks as well: one barrier serves both)
    const u32 oblk = (wg * SETS + (u32)(set ^ 1)) * bpw + (u32)lane;
    const u32 on = oblk < a.nblocks && (u32)lane < bpw ? ((const SCALCE_GLOBAL AcBlockDesc *)a.desc + oblk)->n : 0u;
    const u32 onr = (on + ACL_STEPS - 1) / ACL_STEPS;
    nr_wg = onr > nr_wg ? onr : nr_wg;
  }
#pragma unroll
  for (int d = 32; d; d >>= 1) { const u32 o = __shfl_xor(nr_wg, d, 64); nr_wg = o > nr_wg ? o : nr_wg; }
  nr_wg = __builtin_amdgcn_readfirstlane(nr_wg);
  if (!nr_wg) return;
  // Front stages of other shards run beside this kernel.  Their waves on a coder wave's SIMD take issue slots from it
  // (every one of the four waves is on the round's critical path, the one-instruction-per-four-cycles kind), and their
  // memory traffic queues in front of the gather wave's table rows in the CU's own memory pipeline.  EXCLUSIVE makes each
  // coder wave claim its SIMD's whole register file (256 VGPRs + 256 AGPRs: an empty asm statement that names the last
  // of each; half of it with two sets), so that the dispatcher places nothing else on the CU; otherwise all only run at
  // raised priority.
  if (EXCLUSIVE) {
    if (SETS == 1) asm volatile("" ::: "v255", "a255");
    else asm volatile("" ::: "v255");
  }
  if (SETS == 2 && (role & 1)) {  // the light waves take what the chain / sink on their SIMD leaves
    if (a.helper_prio == 0) __builtin_amdgcn_s_setprio(0);
    else if (a.helper_prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.helper_prio == 3) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(2);
  } else __builtin_amdgcn_s_setprio(3);
  u64 *const prof = a.prof ? a.prof + (size_t)(blockIdx.x * SETS + set) * 5 : nullptr;
  if (prof && lane == 0) atomicOr((unsigned int *)&prof[2], (simd_key() & 3u) << (4 * role));  // profiling: which SIMD each role runs on
  // lanes beyond LW (SETS == 2) hold no block and must not touch the arrays, whose rows are LW wide
  const bool inrow = LW == 64 || lane < LW;

  if (role == 1) {
    // ================= gather: table rows straight into LDS, SLOTS - 1 rounds ahead =================
    constexpr int AHEAD = SLOTS - 1;
    const SCALCE_GLOBAL u8 *sp = (const SCALCE_GLOBAL u8 *)dp->sym;
    const SCALCE_GLOBAL u64 *tab = (const SCALCE_GLOBAL u64 *)dp->tab;  // compact: [6400][81] bounds (ac_table_k)
    // 16 symbols of round k.  The block's symbols are 16-byte aligned (blocks start at multiples of 10 MiB of a 16-byte
    // aligned stream); a read that starts inside the block may run up to 15 bytes past its end (the stream buffers are
    // padded), one that would start past it falls back to the block's first symbols -- garbage nobody uses either way.
    auto sym_addr = [&](u32 k) -> const SCALCE_GLOBAL u8 * {
      const u32 off = k * ACL_STEPS < n ? k * ACL_STEPS : 0u;
      return sp + off;
    };
    u32 p0 = 0, p1 = 0;  // the two symbols in front of the next round to be addressed
    // 16 LDS-direct loads: bounds c and c + 1 of context (p0, p1) in the lane's table -> ops[k % SLOTS][j][lane]
    auto request = [&](const u32x4 sy, u32 k) {
      uint4 *slot = &sh.ops[k % SLOTS][0][0];
#pragma unroll
      for (int j = 0; j < ACL_STEPS; j++) {
        const u32 word = j < 4 ? sy.x : j < 8 ? sy.y : j < 12 ? sy.z : sy.w;
        u32 c = (word >> (8 * (j & 3))) & 0xFFu;
        const u32 D1 = AC_D - 1;
        c = c < D1 ? c : D1;  // symbols >= AC_D raised E_SYMBOL at ingest; stay inside the table regardless
        const u32 idx = (p0 * AC_D + p1) * (AC_D + 1) + c;
        __builtin_amdgcn_global_load_lds((const SCALCE_GLOBAL void *)(tab + idx), (__attribute__((address_space(3))) void *)(slot + j * LW), 16, 0, 0);
        p0 = p1;
        p1 = c;
      }
    };
    // Symbols come 128 bytes per lane at a time (8 rounds; two chunks alternate: A = rounds 0..7 of every 16, B = 8..15),
    // each chunk requested eight rounds before its first symbol is used.  Sixteen bytes per round -- as this wave first
    // did -- touches every 128-byte line of the block eight times, 0.75 us apart, and beside another shard's front stages
    // the line has left the L2 in between: the stream was fetched from HBM several times over and a launch took 1.7 x
    // as long beside the ingest stage as alone.
    // VMEM in issue order: prologue [A B] wait [R0 .. R(AHEAD-1)], then per iteration i [R(i+AHEAD)] and, twice in 16
    // iterations, a chunk of 8 loads behind it; R = 16 instructions.  The hardware counts at most 63 outstanding vector
    // memory instructions per wave: three requests in flight behind the one awaited is as deep as it goes, and everything
    // older than those -- the chunks, requested 8 rounds ahead -- has landed by then.
    u32x4 ca[8], cb[8];
    auto load_chunk = [&](u32x4 (&c)[8], u32 first) {
#pragma unroll
      for (int q = 0; q < 8; q++) c[q] = acl_load16(sym_addr(first + q));
    };
    u64 gprof_wait = 0;
    if (inrow) {
      load_chunk(ca, 0);
      load_chunk(cb, 8);
#pragma unroll
      for (int q = 0; q < 8; q++) { acl_wait_vm<0>(ca[q]); acl_wait_vm<0>(cb[q]); }
#pragma unroll
      for (int q = 0; q < AHEAD; q++) request(ca[q], q);
      acl_wait_vm<(AHEAD - 1) * 16>();  // round 0 is in LDS
      asm volatile("s_barrier" ::: "memory");
      // iteration i: request round i + AHEAD; round i + 1 is in LDS when at most the AHEAD - 1 requests behind it are outstanding
      auto close = [&]() {
        acl_wait_vm<(AHEAD - 1) * 16>();
        if (prof) {
          const u64 w0 = __builtin_amdgcn_s_memtime();
          asm volatile("s_barrier" ::: "memory");
          gprof_wait += __builtin_amdgcn_s_memtime() - w0;
        } else {
          asm volatile("s_barrier" ::: "memory");
        }
      };
      for (u32 i0 = 0; i0 < nr_wg; i0 += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          if (i0 + j < nr_wg) {
            // round i0 + j + AHEAD: its symbols are element (j + AHEAD) % 16 of the chunk pair; chunk A is free for the rounds
            // 16 .. 23 behind i0 once round i0 + 7 has been requested (j = 8 - AHEAD), chunk B eight iterations later
            const int e = (j + AHEAD) & 15;
            request(e < 8 ? ca[e] : cb[e - 8], i0 + j + AHEAD);
            if (j == 8 - AHEAD) load_chunk(ca, i0 + 16);
            if (j == 16 - AHEAD) load_chunk(cb, i0 + 24);
            close();
          }
        }
      }
      acl_wait_vm<0>();  // nothing of this wave may land in LDS after the workgroup has gone
      asm volatile("s_barrier" ::: "memory");  // (the sink's last round)
      asm volatile("s_barrier" ::: "memory");  // (the writer's last words)
    }
    if (prof && lane == 0) prof[3] = gprof_wait;
  } else if (role == 0) {
    // ================= chain: one coder state per lane =================
    u32 ones, zero;
    asm("v_mov_b32 %0, -1" : "=v"(ones));
    asm("v_mov_b32 %0, 0" : "=v"(zero));
    u32 lo = 0, M = 0;  // M = 0 stands for 2^32
    u32 Q = 16;         // stream position of the next B's top bit: the two raw symbols, then every bit dropped
    u64 prof_wait = 0;
    const u64 prof_t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
    if (inrow) {
      __syncthreads();  // round 0's operands are in LDS
      for (u32 r = 0; r < nr_wg; r++) {
        const int slot = r % SLOTS;
        const u32 lo0 = lo, M0 = M, Q0 = Q;
        // all 16 operand reads are issued before the first step (read at the point of use, every step waited for a full
        // LDS round trip)
        uint4 g[ACL_STEPS];
#pragma unroll
        for (int j = 0; j < ACL_STEPS; j++) g[j] = sh.ops[slot][j][lane];
        u32 topw = 0;  // g(c_hi) = 2^64 - 1 marks the last symbol of a context (ac_table_k); no regular high word reaches that
        u32 Bp = 0, Qp = 0;  // (the even step of a pair: two steps leave in one 16-byte record)
#pragma unroll
        for (int j = 0; j < ACL_STEPS; j++) {
          u32 B, t;
          acl_step_plain(lo, M, g[j], ones, zero, B, t);
          Q += t;
          if (j & 1) sh.rec[r & 1][j >> 1][lane] = make_uint4(Qp, Bp, Q, B);  // (Q, B): B is the HIGH word of its product's register pair -- an odd register, as .y and .w are; the other order cost two v_mov per step
          else { Bp = B; Qp = Q; }
          topw = g[j].w > topw ? g[j].w : topw;
        }
        const bool live = r < nr;
        const bool poisoned = a.test_poison && r % a.test_poison == 0;
        const bool complete = r > 0 && r * ACL_STEPS + ACL_STEPS <= n;  // not the round of the raw symbols, not a tail
        const bool fast_ok = complete && topw != 0xFFFFFFFFu && M0 != 0u && M != 0u && !poisoned;
        const bool fix = live && !fast_ok;  // this lane's records of the round are rewritten by general steps (a lane whose
                                            // block has ended computes garbage nobody reads: the sink skips it)
        if (__builtin_expect(__any(fix), 0)) {
          if (fix) {
            u32 glo = lo0 & 0x7FFFFFFFu, ghi = glo + M0 - 1u, gq = Q0;
            const u32 jstart = r == 0 ? 2u : 0u;
            const u32 left = n - r * ACL_STEPS;
            const u32 jend = left < (u32)ACL_STEPS ? left : (u32)ACL_STEPS;
            u32 Bp = 0, Qp = 0;
#pragma unroll 1
            for (u32 j = 0; j < (u32)ACL_STEPS; j++) {
              const uint4 gj = sh.ops[slot][j][lane];
              u32 B = 0, t = 0;
              if (j >= jstart && j < jend) acl_step_general(glo, ghi, gj, B, t);
              gq += t;
              if (j & 1u) sh.rec[r & 1][j >> 1][lane] = make_uint4(Qp, Bp, gq, B);
              else { Bp = B; Qp = gq; }
            }
            lo = glo;
            M = ghi - glo + 1u;
            Q = gq;
          }
        }
        if (r + 1 == nr) sh.final_lo[lane] = lo & 0x7FFFFFFFu;
        if (prof) {
          const u64 w0 = __builtin_amdgcn_s_memtime();
          __syncthreads();
          prof_wait += __builtin_amdgcn_s_memtime() - w0;
        } else {
          __syncthreads();
        }
      }
      __syncthreads();  // (the sink's last round)
      __syncthreads();  // (the writer's last words)
    }
    if (prof && lane == 0) {
      prof[0] = prof_wait;
      prof[1] = __builtin_amdgcn_s_memtime() - prof_t0;
    }
  } else if (role == 2) {
    // ================= sink: one bit accumulator per lane, one round behind the chain =================
    u64 sprof_wait = 0;
    if (inrow) {
      AclSink<LW> sk;
      {
        const SCALCE_GLOBAL u8 *sp = (const SCALCE_GLOBAL u8 *)dp->sym;
        const u32 s0 = n ? (u32)sp[0] : 0u, s1 = n > 1 ? (u32)sp[1] : 0u;
        sk.init((SCALCE_GLOBAL u32 *)dp->dst, dp->cap / 4, (SCALCE_GLOBAL u32 *)dp->log, dp->log_cap, s0, s1);
      }
      sh.pub[0][lane] = 0;
      sh.pub[1][lane] = 0;
      barrier_lds_only();
      auto take = [&](u32 r) {  // the records of round r
        if (r < nr && sk.room()) {  // (per lane: a block that has ended has no records)
          uint4 v[ACL_STEPS / 2];
#pragma unroll
          for (int j = 0; j < ACL_STEPS / 2; j++) v[j] = sh.rec[r & 1][j][lane];
          const u32 s2 = sk.w2, s1 = sk.w1, s0 = sk.w0, sq = sk.Qb;
          sk.ncar = ~0u;
          // two steps, one record for the writer wave: the ring stores (an LDS instruction and its address per step) are its
          // work, not this wave's -- the sink sets the pace of a round
#pragma unroll
          for (int j = 0; j < ACL_STEPS / 2; j++) {
            uint4 o;
            sk.step(v[j].y, v[j].x, o.x, o.y);
            sk.step(v[j].w, v[j].z, o.z, o.w);
            sh.fifo[r & 1][j][lane] = o;
          }
          if (__builtin_expect(__any(sk.ncar == 0u), 0)) {
            if (sk.ncar == 0u) {
              sk.w2 = s2; sk.w1 = s1; sk.w0 = s0; sk.Qb = sq;
#pragma unroll 1
              for (int j = 0; j < ACL_STEPS / 2; j++) {
                const uint4 x = sh.rec[r & 1][j][lane];
                sk.careful(x.y, x.x);
                sk.careful(x.w, x.z);
              }
            }
          }
        }
      };
      for (u32 i = 0; i < nr_wg; i++) {  // iteration i: round i - 1
        if (i > 0) take(i - 1);
        sh.pub[i & 1][lane] = sk.final_words();
        if (prof) {
          const u64 w0 = __builtin_amdgcn_s_memtime();
          barrier_lds_only();
          sprof_wait += __builtin_amdgcn_s_memtime() - w0;
        } else {
          barrier_lds_only();
        }
      }
      take(nr_wg - 1);
      sh.pub[nr_wg & 1][lane] = sk.final_words();
      barrier_lds_only();   // the writer takes out every final word ...
      barrier_lds_only();   // ... and they are in memory (its fence): the notes go on top
      if (have && n) {
        const u32 bytes = sk.finish(sh.final_lo[lane]);
        *(SCALCE_GLOBAL u32 *)dp->out_size = sk.over ? 0u : bytes;  // (a block that ran out of room: nothing frames bytes it does not hold)
        if (sk.over) dev_fail(dp->err, E_ACOVERFLOW, dp->index, bytes);
      }
    }
    if (prof && lane == 0) prof[4] = sprof_wait;
  } else {
    // ================= writer: final words out of the staging ring =================
    constexpr u32 RING = AclRing<LW, SETS>::WORDS;
    SCALCE_GLOBAL u32 *dst = (SCALCE_GLOBAL u32 *)dp->dst;
    const u32 wcap = dp->cap / 4;
    u32 wo = 0;  // words of this lane's block in global memory (multiple of 32 until the end)
    // In place (AC_BLOCK_IN_PLACE: dst IS the block's symbols): a line may only go where the gather wave has been for good.  In
    // iteration i that wave is requesting round i + SLOTS - 1 out of chunks of eight rounds it loaded eight rounds earlier; what
    // is in flight lies behind round i: lines end in front of round i - 1 (4 words per round).  A line that is ready and may
    // not go is the end of the block -- its output has caught up with its input: E_ACOVERFLOW, nothing more is written (the
    // host runs the shard again with buffers of its own).
    const bool in_place = have && (dp->flags & AC_BLOCK_IN_PLACE) != 0;
    bool dead = false;
    auto word = [&](u32 k) -> u32 { return __builtin_bswap32(sh.stage[(k + 1u) & (RING - 1)][lane]); };
    // Whole 128-byte lines, the lines of TWO blocks per visit: threads 0..31 store the 32 words of one block's line, threads
    // 32..63 those of another's -- one coalesced 128-byte store each, and a visit per pair of lines that ARE ready.  (Round 3:
    // every lane stored its own line in eight 16-byte pieces, all lanes masked but the ready ones -- 64 scattered pieces per
    // instruction, and the visit cost the same ~1900 cycles for one ready lane as for forty; with some lane ready in nearly
    // every round it was half of this wave's time.)
    auto drain = [&](u32 lim, u32 words_taken) {
      if (in_place && !dead && lim >= wo + 32u && wo + 32u > words_taken) {
        dead = true;
        dev_fail(dp->err, E_ACOVERFLOW, dp->index, 4u * (wo + 32u));
      }
      u64 ready = __ballot(inrow && !dead && lim >= wo + 32u);
      while (ready) {
        const u32 la = (u32)__builtin_ctzll(ready);
        ready &= ready - 1;
        const u32 lb = ready ? (u32)__builtin_ctzll(ready) : la;  // (one line left: the upper threads stand by)
        const bool two = ready != 0;
        ready &= ready - 1;
        const u32 t = (u32)lane & 31u;
        const bool up = lane >= 32;
        const u32 src = up ? lb : la;
        const u32 wo_a = (u32)__builtin_amdgcn_readlane(wo, la), wo_b = (u32)__builtin_amdgcn_readlane(wo, lb);
        const u32 cap_a = (u32)__builtin_amdgcn_readlane(wcap, la), cap_b = (u32)__builtin_amdgcn_readlane(wcap, lb);
        // (the builtin returns a signed int: without the casts a low word with bit 31 set sign-extends into the high word)
        const u64 da = (u64)(uintptr_t)dst;
        const u32 dlo = (u32)da, dhi = (u32)(da >> 32);
        const u64 dst_a = ((u64)(u32)__builtin_amdgcn_readlane(dhi, la) << 32) | (u64)(u32)__builtin_amdgcn_readlane(dlo, la);
        const u64 dst_b = ((u64)(u32)__builtin_amdgcn_readlane(dhi, lb) << 32) | (u64)(u32)__builtin_amdgcn_readlane(dlo, lb);
        const u32 k = (up ? wo_b : wo_a) + t;
        const u32 v = __builtin_bswap32(sh.stage[(k + 1u) & (RING - 1)][src]);
        SCALCE_GLOBAL u32 *d = (SCALCE_GLOBAL u32 *)(uintptr_t)(up ? dst_b : dst_a);
        if ((!up || two) && k < (up ? cap_b : cap_a)) d[k] = v;
        if ((u32)lane == la || (two && (u32)lane == lb)) wo += 32;
      }
    };
    // the words the sink's steps of round r reported (AclSink::step), into the ring in step order: a later report of the
    // same word replaces an earlier one.  Only for a round the lane's block has -- what lies in the records otherwise is an
    // older round's, and putting that back would undo newer words that are not out yet.
    auto apply = [&](u32 r) {
      if (inrow && r < nr) {
        uint4 e[ACL_STEPS / 2];
#pragma unroll
        for (int j = 0; j < ACL_STEPS / 2; j++) e[j] = sh.fifo[r & 1][j][lane];
#pragma unroll
        for (int j = 0; j < ACL_STEPS / 2; j++) {
          sh.stage[e[j].y & (RING - 1)][lane] = e[j].x;
          sh.stage[e[j].w & (RING - 1)][lane] = e[j].z;
        }
      }
    };
    barrier_lds_only();
    for (u32 i = 0; i < nr_wg; i++) {  // iteration i: the records of round i - 2 (the sink took it in iteration i - 1), then what it published there
      if (i > 1) apply(i - 2);
      if (i > 0) drain(inrow ? sh.pub[(i - 1) & 1][lane] : 0u, i > 1 ? (4u * (i - 1u)) >> a.inplace_shift : 0u);
      barrier_lds_only();
    }
    barrier_lds_only();  // the sink's last round is in the records
    {
      if (nr_wg > 1) apply(nr_wg - 2);
      apply(nr_wg - 1);
      const u32 lim = inrow ? sh.pub[nr_wg & 1][lane] : 0u;
      drain(lim, ~0u);   // (every symbol has been taken)
      for (; wo < lim && !dead; wo++)
        if (wo < wcap) dst[wo] = word(wo);
      __threadfence();
    }
    barrier_lds_only();
  }
}

}  // namespace scalce
