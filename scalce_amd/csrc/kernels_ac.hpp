// kernels_ac.hpp -- static order-2 arithmetic coder over the reordered quality stream.
// Reference: ac_stat (/root/reference/arithmetic.cpp:54-78), ac_coder::{reset,O,write,flush}
// (:85-169), block framing of ac_write/thread_c (:280-287,318-363), table scaling
// (compress.cpp:296-320), decoder (:173-268).
//
// Parallelism is the reference's own: independent 10 MiB blocks.  Inside a block the coder state
// (lo, hi, underflow) is a serial chain, so one 64-lane wavefront owns one block:
//   * all 64 lanes fetch the next 64 symbols, form their contexts and gather the two cumulative
//     bounds of each symbol in parallel (the table lookups do not depend on the coder state);
//   * the bounds are stored as 64-bit reciprocal fractions g = floor(c * 2^64 / total) + 1, so
//     the two 64/32-bit divisions of arithmetic.cpp:131-132 become multiply-high:
//       floor(range * c / total) == floor(range * g / 2^64)   for range <= 2^32, c < total < 2^32
//     (error of g is at most 2^-64, times range at most 2^-32 < 1/total, so no integer is crossed);
//   * the serial part then walks the 64 staged operands from LDS (broadcast reads) with the
//     renormalisation loop of :133-152 in closed form (count-leading-zeros instead of bit steps).
#pragma once
#include <type_traits>
#include "kernels_order.hpp"

namespace scalce {

constexpr int AC_D = 80;
constexpr u32 AC_BLOCK_SYMS = 10u * 1024u * 1024u;

// scaled table: max(1, (1 + count) / factor)  (all counters start at 1, qualities.cpp:191-196;
// compress.cpp:310-313)
__global__ __launch_bounds__(256) void ac_scale_k(const u64 *freq4, u32 factor, u32 *table) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= AC_D * AC_D * AC_D) return;
  u64 p = (freq4[i] + 1) / factor;
  table[i] = p ? (u32)p : 1u;
}

// per (context, symbol): {g(lo) low, g(lo) high, g(hi) low, g(hi) high}.  c_hi == total (the last symbol of
// a context) is marked by g(hi) = 2^64-1: a regular g is at most 2^64 - 2^64/total + 1 < 2^64 - 2^32, so its
// high word never reaches 0xFFFFFFFF.
__device__ __forceinline__ u64 recip_frac(u32 c, u32 d) {
  const u64 qh = ((u64)c << 32) / d;
  const u64 rem = ((u64)c << 32) - qh * d;
  const u64 ql = (rem << 32) / d;
  return ((qh << 32) | ql) + 1;
}
// `tab8` (may be null): the same fractions as one u64 per cumulative bound, [6400][81] -- g(c_hi) of a symbol is g(c_lo) of
// the next, so the 16 bytes at &tab8[ctx * 81 + s] ARE symbol s's operands and the table is half the size (4.1 MB): the
// one-block-per-lane coder gathers from it (kernels_acl.hpp; an XCD's 4 MB L2 holds the rows in use of one table, not of two).
// cost[0] += sum over the context's symbols of f * log2(tot / f) in 1/256 bit, cost[1] += tot: what coding the table's own
// counts with the table costs -- the host sizes the coder's block buffers from it (ac_prepare)
__global__ __launch_bounds__(64) void ac_table_k(const u32 *table, uint4 *tab, u32 *cum /*[6400][81]*/, u32 *max_total, u64 *tab8 = nullptr,
                                                unsigned long long *cost = nullptr) {
  const u32 ctx = blockIdx.x * blockDim.x + threadIdx.x;
  if (ctx >= AC_D * AC_D) return;
  const u32 *f = table + (u64)ctx * AC_D;
  u32 tot = 0;
  for (int s = 0; s < AC_D; s++) tot += f[s];
  if (max_total) atomicMax(max_total, tot);
  if (cost) {
    const float lt = __log2f((float)tot);
    float bits = 0.f;
    for (int s = 0; s < AC_D; s++) bits += (float)f[s] * (lt - __log2f((float)f[s]));
    atomicAdd(&cost[0], (unsigned long long)(bits * 256.f));
    atomicAdd(&cost[1], (unsigned long long)tot);
  }
  u32 run = 0;
  u64 glo = 1;  // c == 0 -> quotient 0
  cum[ctx * 81] = 0;
  if (tab8) tab8[(u64)ctx * 81] = glo;
  for (int s = 0; s < AC_D; s++) {
    run += f[s];
    cum[ctx * 81 + s + 1] = run;
    const u64 ghi = (run == tot) ? ~0ull : recip_frac(run, tot);
    tab[(u64)ctx * AC_D + s] = make_uint4((u32)glo, (u32)(glo >> 32), (u32)ghi, (u32)(ghi >> 32));
    if (tab8) tab8[(u64)ctx * 81 + s + 1] = ghi;
    glo = ghi;
  }
}

// one 10 MiB block of a coder launch that spans several symbol streams (ac_encode_rows_k)
struct AcBlockDesc {
  const u8 *sym;        // first symbol of the block
  const uint4 *tab;     // reciprocal-fraction table of the block's stream
  u32 *dst;             // output words of the block
  u32 *out_size;        // where its byte count goes
  DevErr *err;          // error word of the shard it belongs to
  u32 n;                // symbols in the block
  u32 index;            // block index inside its stream (error reports)
  u32 cap;              // bytes the block may write at dst (multiple of 4)
  u32 flags;            // AC_BLOCK_IN_PLACE: dst IS the block's symbols (scalce_batch_set_code_in_place) -- the coded bytes go over
                        // what the coder has already consumed, and a block whose output would catch up with its input reports
                        // E_ACOVERFLOW instead
  u32 *log;             // ac_encode_lanes_k: the block's carry notes (AclSink), `log_cap` words; null: at the end of dst
  u32 log_cap;
  u32 pad_;
};
constexpr u32 AC_BLOCK_IN_PLACE = 1u;
struct AcEncArgs {
  const u8 *sym;
  u64 nsym;
  const uint4 *tab;
  u8 *out;          // block b writes at out + b * out_stride (multiple of 4)
  u64 out_stride;
  u32 out_cap;      // bytes a block may write (multiple of 4)
  u32 *out_size;
  DevErr *err;
  u32 slow_threshold;  // 32; tests lower it to drive every pending underflow through the serial path
  u32 *simd_load;      // [AC_SIMD_KEYS] coder waves per SIMD of the device, shared by every launch (may be null)
  u64 *prof;           // profiling only (SCALCE_AC_PROF): per block {cycles in the 64 steps, cycles in the rest of the round, rounds}
  const AcBlockDesc *desc;  // ac_encode_rows_k: one entry per block of the launch
  u32 nblocks;
  u32 chain_prio;      // s_setprio of the chain waves (ac_encode_rows_k)
  u32 helper_prio;     // s_setprio of the helper waves (ac_encode_rows_k)
  u32 lanes_used;      // ac_encode_lanes_k: blocks per workgroup (0 = 64)
  u32 pairing;         // ac_encode_lanes_k with two sets: which roles share a SIMD (experiments)
  u32 test_poison;     // test hook: every test_poison-th super-round pretends a step hit the full-range exit (0 = off)
  u32 inplace_shift;   // test hook (SCALCE_AC_INPLACE_TEST): blocks coded in place may only use 2^-shift of what they have consumed
};
// index of the SIMD a wave runs on: XCC_ID[3:0] | HW_ID{se_id, sh_id, cu_id}[15:8] | HW_ID simd_id[5:4]
constexpr u32 AC_SIMD_KEYS = 16u << 10;
__device__ __forceinline__ u32 simd_key() {
  u32 hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 15u) << 10) | (((hw >> 8) & 0xFFu) << 2) | ((hw >> 4) & 3u);
}

// floor((R + 1) * g / 2^64) for R < 2^32, g < 2^64 -- plain form, used by the self-test as the yardstick
__device__ __forceinline__ u32 mulfrac(u32 R, u32 g_lo, u32 g_hi) {
  const u64 t0 = (u64)R * g_lo + g_lo;
  const u64 t1 = (u64)R * g_hi + g_hi + (t0 >> 32);
  return (u32)(t1 >> 32);
}
// same value with M = R + 1 taken modulo 2^32 and `wrap` = (R == 0xFFFFFFFF): three instructions
// (v_mul_hi_u32, v_cndmask, v_mad_u64_u32).  When M wrapped to 0 the product term vanishes and the
// missing 2^32 * g is put back through the high word of the addend.
__device__ __forceinline__ u32 mulfrac_m(u32 M, bool wrap, u32 g_lo, u32 g_hi) {
  const u64 add = ((u64)(wrap ? g_hi : 0u) << 32) | __umulhi(M, g_lo);
  return (u32)(((u64)M * g_hi + add) >> 32);
}

// One coder step (arithmetic.cpp:122-152) in closed form.  Returns k | u << 8; `hbefore` = hi before the
// shift (its top k bits are the bits to emit).
//   k = leading bits on which lo and hi agree (:134-139), u = the following positions where lo has 1 and
//   hi has 0 ("underflow ante portas", :140-146); the loop can only run k steps of the first kind and
//   then u of the second.
// GENERAL = false is the production path: it assumes a well-formed coder state (lo <= hi, lo = 0..,
// hi = 1.. after every step), which holds whenever no context total exceeds 2^30 -- then every symbol
// keeps a non-empty interval because range > 2^30 >= total.  Both shifts are merged into one, and shifting
// by min(k, 31) also gives the right state (lo = 0, hi = ~0) when all 32 bits agree.
// GENERAL = true reproduces the reference bit for bit on ANY state, including the inverted intervals
// (hi < lo) its 32-bit arithmetic runs into once a context total passes 2^30; the host selects it from the
// table (scalce_batch_entropy).
template <bool GENERAL>
__device__ __forceinline__ u32 ac_step(u32 &lo, u32 &hi, const uint4 g, u32 &hbefore) {
  const u32 R = hi - lo;
  u32 M;
  const bool wrap = __builtin_add_overflow(R, 1u, &M);
  const u32 qa = mulfrac_m(M, wrap, g.z, g.w);
  const u32 qb = mulfrac_m(M, wrap, g.x, g.y);
  const u32 nhi = (g.w == 0xFFFFFFFFu) ? hi : lo + qa - 1;  // c_hi == total: hi unchanged
  const u32 nlo = lo + qb;
  hbefore = nhi;
  const u32 x = nlo ^ nhi;
  const u32 kf = x ? (u32)__clz(x) : 0xFFFFFFFFu;
  const u32 krec = kf < 32u ? kf : 32u;
  if (!GENERAL) {
    const u32 ks = kf < 31u ? kf : 31u;
    const u32 z = ((nlo & ~nhi) << ks) << 1;
    const u32 u = (u32)__clz(~z);  // z has bit 0 clear, so ~z != 0 and u <= 31 - ks
    const u32 t = ks + u;
    lo = (nlo << t) & 0x7FFFFFFFu;
    hi = (nhi << t) | ((1u << t) - 1) | 0x80000000u;
    return krec | (u << 8);
  } else {
    u32 l1, h1;
    if (krec == 32) { l1 = 0; h1 = 0xFFFFFFFFu; }
    else { l1 = nlo << krec; h1 = (nhi << krec) | ((1u << krec) - 1); }
    const u32 y = (l1 & ~h1) << 1;
    const u32 u = (u32)__clz(~y);
    lo = u ? ((l1 << u) & 0x7FFFFFFFu) : l1;
    hi = u ? ((h1 << u) | ((1u << u) - 1) | 0x80000000u) : h1;
    return krec | (u << 8);
  }
}

// The step of a "plain" round, on the state (lo, M) with M = hi - lo + 1 taken modulo 2^32.  Renormalising
// by t bits multiplies the range by 2^t whichever mix of agreeing-bit and underflow steps t is made of, so
// M' = (A - B) << t and hi never has to be rebuilt.  Preconditions, checked by the caller: no symbol of the
// round is the last one of its context (helper wave), and M != 0, i.e. the interval is not the full 2^32.
// Two outcomes need the general step and are reported through the return value being 0: all 32 bits agree
// (x == 0), or the new interval is an aligned power-of-two block that renormalises to the full 2^32
// (M' == 0).  In that case lo and M are left untouched and the caller redoes the symbol.
__device__ __forceinline__ u32 ffbh_raw(u32 x) {  // count leading zeros, 0xFFFFFFFF for 0 (callers discard that case)
  u32 r;
  asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// Returns e (per lane; 0 iff this step needs the general path).  The state is always advanced: the caller
// checks the minimum of e over a short group of steps with ONE scalar branch and, if it is 0, rolls the
// group back and redoes it with the general step -- a branch per symbol would sit on the dependency spine
// (the wave cannot issue past an unresolved branch).  After a bad step the following plain steps of the
// group compute garbage, but only with plain integer operations.
__device__ __forceinline__ u32 ac_step_plain(u32 &lo, u32 &M, const uint4 g, u32 &hbefore, u32 &ku) {
  // dependency spine: M -> mul_hi -> mad -> add -> xor -> ffbh -> shift -> ffbh -> shift -> M'
  const u32 A = (u32)(((u64)M * g.w + __umulhi(M, g.z)) >> 32);
  const u32 B = (u32)(((u64)M * g.y + __umulhi(M, g.x)) >> 32);
  const u32 nhi = lo + A - 1;
  const u32 nlo = lo + B;
  const u32 W = A - B;                       // new range before renormalisation (off the spine)
  hbefore = nhi;
  const u32 x = nlo ^ nhi;
  const u32 k = ffbh_raw(x);                 // x == 0 leaves garbage here; that case is redone by the caller
  // underflow steps = leading ones of ((nlo & ~nhi) << k) << 1.  Counted on the complement so that no
  // inversion sits behind the shift: c1 = ~((nlo & ~nhi) << 1) has bit 0 set, and the zeros shifted into
  // (c1 << k) lie below the first set bit (u <= 31 - k).
  const u32 c1 = ((~nlo | nhi) << 1) | 1u;
  const u32 u = ffbh_raw(c1 << k);
  M = (W << k) << u;                         // renormalising by k + u bits scales the range by 2^(k+u)
  lo = ((nlo << k) << u) & 0x7FFFFFFFu;
  ku = k | (u << 8);
  return x < M ? x : M;                      // 0 iff x == 0 or the range wrapped to 2^32
}

// ---- systolic plain round ----------------------------------------------------------------------
// 64 plain steps (see ac_step_plain) of one round.  Lane s holds the operands of symbol s; the state (lo, M)
// walks one lane per step with DPP wave_shr:1, so operands never move and nothing is broadcast.  Every lane
// executes every step, only step s is meaningful for lane s.  A lone wave issues one instruction per 4 cycles
// whether it depends on the previous one or not (tools/ubench_issue2.hip), so the cost of a symbol is the
// number of instructions in a step and nothing else.  Hence:
//   * no outcome leaves the loop.  Each lane only keeps the state it received in ITS step; after the loop all
//     64 lanes redo their own symbol at once (k, u, hi-before-shift, exit test).
//   * keeping costs nothing: the DPP moves that bring the state in are restricted to the quad of lane s by
//     their row_mask / bank_mask (write enables per 16 and per 4 lanes), and consecutive steps write four
//     rotating register sets, so what lane s received in step s is never overwritten.
//   * lo travels without its bit 31 cleared ((nlo << t) & 0x7FFFFFFF in ac_step_plain): a stray bit 31 flips
//     bit 31 of nlo and nhi alike, which changes neither x = nlo ^ nhi nor the underflow count (its mask is
//     shifted left by one), and is shifted out or carried as is.  The caller removes it afterwards.
// Per step: dpp mov, 2 x (mul_hi, mad), sub, dpp add, add3, xor, ffbh, bitop, lshl_or, lshl, ffbh, add, 2 x lshl.
struct SysState {
  u32 kM[4], nl[4];  // per lane: range received / lo + B computed in the lane's own step (set = lane & 3)
  u32 ones;          // 0xFFFFFFFF
  u32 zero;          // 0, just as opaque: a literal 0 as the high word of B's addend is re-created in the result's
                     // register pair before every multiply-add (one v_mov per step)
};
// Renormalisation count without the (k, u) pair.  The loop of arithmetic.cpp:133-152 drops t = k + u leading bits,
// and t is the largest number of halvings after which [nlo, nhi] still lies inside ONE window of the form
// [m h, m h + 2 h), h = 2^(31 - t) (same top bit: window at an even m; 01.. / 10..: window at an odd m).  With
// D = nhi - nlo >= 1 and D's top bit at position r, no window narrower than 2^(r+1) holds the interval, and the one
// of exactly that width does unless adding D to the lower r bits of nlo reaches 2^(r+1):
//     t = clz( (nlo mod 2^r) + D ),    r = 31 - clz(D)
// (checked against the literal loop on random and crafted intervals in tests/test_host_cpu.py, and end to end by
// every parity test).  Five instructions -- ffbh, not, bfe, add, ffbh -- instead of seven, hi is never formed.
__device__ __forceinline__ u32 renorm_count(u32 nlo, u32 D) {
  const u32 c = (u32)__builtin_clz(D);                    // D == 0: garbage, found by the caller's exit test
  const u32 low = __builtin_amdgcn_ubfe(nlo, 0u, ~c);    // width = (31 - c) in the 5 bits the instruction reads
  return (u32)__builtin_clz(low + D);
}
template <int S>
__device__ __forceinline__ void sys_step(SysState &st, u32 &tlo, u32 &tM, const uint4 &ops) {
  constexpr int Q = S & 3;
  constexpr int RM = 1 << (S >> 4), BM = 1 << ((S & 15) >> 2);
  // lane l takes the state lane l-1 produced in the previous step; only the quad of lane S is written
  st.kM[Q] = __builtin_amdgcn_update_dpp(st.kM[Q], tM, 0x138, RM, BM, false);
  const u32 M = st.kM[Q];
  // A - 1 comes for free: the high word of the 64-bit addend is 2^32 - 1 (st.ones: opaque to the compiler, which
  // would otherwise pull the constant out of the multiply-add and spend an instruction on it)
  const u32 A1 = (u32)(((u64)M * ops.w + (((u64)st.ones << 32) | __umulhi(M, ops.z))) >> 32);
  const u32 B = (u32)(((u64)M * ops.y + (((u64)st.zero << 32) | __umulhi(M, ops.x))) >> 32);
  const u32 D = A1 - B;  // new range - 1 = nhi - nlo
  // nl = lo(from lane l-1) + B in one instruction; tlo was written many instructions ago (DPP read hazard)
  asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:%3 bank_mask:%4"
      : "+v"(st.nl[Q]) : "v"(tlo), "v"(B), "n"(RM), "n"(BM));
  // keep the two instructions of renorm_count that only need D behind the asm statement: the compiler assumes the
  // worst about what an asm statement writes and puts a wait state in front of an instruction that reads its
  // result right away
  __builtin_amdgcn_sched_barrier(0);
  const u32 nlo = st.nl[Q];
  const u32 t = renorm_count(nlo, D);          // D < 2: garbage, found by the caller's exit test
  tM = (D + 1) << t;                           // renormalising by t bits scales the range by 2^t
  tlo = nlo << t;
}
template <int S, int E>
struct SysLoop {
  static __device__ __forceinline__ void run(SysState &st, u32 &tlo, u32 &tM, const uint4 &ops) {
    sys_step<S>(st, tlo, tM, ops);
    SysLoop<S + 1, E>::run(st, tlo, tM, ops);
  }
};
template <int E>
struct SysLoop<E, E> {
  static __device__ __forceinline__ void run(SysState &, u32 &, u32 &, const uint4 &) {}
};
__device__ __forceinline__ void sys_round(SysState &st, u32 lo, u32 M0, const uint4 &ops, u32 &tlo, u32 &tM) {
  // step 0: every lane starts from the round's state (lane 0 is the one that matters); the state after the round is
  // what lane 63 leaves in (tlo, tM)
  asm("v_mov_b32 %0, -1" : "=v"(st.ones));
    asm("v_mov_b32 %0, 0" : "=v"(st.zero));
  {
    st.kM[0] = M0;
    st.kM[1] = st.kM[2] = st.kM[3] = 0;
    const u32 A = (u32)(((u64)M0 * ops.w + __umulhi(M0, ops.z)) >> 32);
    const u32 B = (u32)(((u64)M0 * ops.y + __umulhi(M0, ops.x)) >> 32);
    const u32 W = A - B;
    st.nl[0] = lo + B;
    st.nl[1] = st.nl[2] = st.nl[3] = 0;
    const u32 nlo = st.nl[0];
    const u32 t = renorm_count(nlo, W - 1);
    tM = W << t;
    tlo = nlo << t;
  }
  SysLoop<1, 64>::run(st, tlo, tM, ops);
}

// ---- encoder -----------------------------------------------------------------------------------
// One workgroup of two wavefronts per 10 MiB block, 64 symbols per round:
//   wave 1, gather  (64 lanes)  operands of round r+1: context -> {g(lo), g(hi)} from the table -> LDS
//   wave 0, chain   (1 lane)    the coder state walks the 64 symbols of round r: operands come from LDS,
//                               the per-symbol outcome (hi before the shift, k = agreed leading bits,
//                               u = underflow steps) goes back to LDS.  No bit output on this path.
//                               Measured (tools/ubench_chain.hip): the chain costs ~90 ns per symbol as
//                               vector code against ~145 ns as scalar code, because v_mad_u64_u32 does in
//                               one instruction what takes five scalar ones, and issue is ~5-6 cycles per
//                               dependent instruction either way.
//   wave 1, pack    (64 lanes)  the 64 outcomes of round r-1 become bits at once: a segmented scan
//                               resolves the pending-underflow counts (arithmetic.cpp:136-139), a prefix
//                               sum gives every symbol its bit offset, the bits are OR-ed into an LDS
//                               word buffer and the finished words leave with one coalesced store.
// The two waves meet at one barrier per round, so gather and pack are off the chain's critical path.
constexpr int AC_BUF_WORDS = 200;  // up to 2047 buffered bits (ac_pack2) + 64 symbols x 64 bits, rounded up
constexpr int AC_ROUND_WORDS = 136;  // one round on its own (AcSink::pack): 31 carried bits + 64 symbols x 64 bits

__device__ __forceinline__ void lds_place(u32 *buf, u32 bits, u32 n, u32 bitpos) {  // n in 1..32
  const u32 s = bitpos & 31, w = bitpos >> 5;
  const u64 x = (u64)bits << (64 - s - n);
  atomicOr(&buf[w], (u32)(x >> 32));
  if (s + n > 32) atomicOr(&buf[w + 1], (u32)x);
}

// Bit sink of one block (helper-wave side): takes the per-symbol outcomes of a round -- hi before the shift,
// k agreed leading bits, u underflow steps -- and appends the bits arithmetic.cpp:133-147 would have written one at a
// time.  All 64 lanes of the calling wave take part; `buf` is AC_BUF_WORDS words of LDS owned by that wave.
struct AcSink {
  SCALCE_GLOBAL u32 *dst = nullptr;  // the block's output words
  u32 wcap = 0;        // words the block may write
  u32 gw = 0;          // words already stored
  u32 c0 = 16;         // bits pending in `carry` (left aligned); a block starts with its two raw symbols
  u32 carry = 0;
  u32 pend = 0;        // underflow steps not yet materialised as bits
  bool over = false;
  u32 fill = 0;        // ac_pack2 only: bits waiting in the wave's LDS buffer (stored 2048 at a time)

  // ac_pack2 <-> the one-round-at-a-time form: everything buffered goes out except the last partial word, which becomes
  // `carry` / `c0` again (to_plain), or the partial word goes back to the front of the buffer (to_buffered)
  __device__ __forceinline__ void to_buffered(u32 *buf, int lane) {
    if (lane == 0) buf[0] = carry;
    fill = c0;
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void to_plain(u32 *buf, int lane) {
    const u32 nfull = fill >> 5;
    for (u32 w = lane; w < nfull; w += 64) {
      if (gw + w < wcap) dst[gw + w] = __builtin_bswap32(buf[w]);
      else over = true;
    }
    carry = (fill & 31) ? buf[nfull] : 0u;
    c0 = fill & 31;
    gw += nfull;
    fill = 0;
    __builtin_amdgcn_wave_barrier();
  }

  // uniform append of nb <= 32 bits (every lane passes the same values): rare slow path and final flush
  __device__ __forceinline__ void emit_u(u32 *buf, int lane, u32 v, u32 nb) {
    for (int w = lane; w < 4; w += 64) buf[w] = (w == 0) ? carry : 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) lds_place(buf, v, nb, c0);
    __builtin_amdgcn_wave_barrier();
    const u32 endbits = c0 + nb;
    const u32 nfull = endbits >> 5;  // 0 or 1
    if (nfull && lane == 0) {
      if (gw < wcap) dst[gw] = __builtin_bswap32(buf[0]);
      else over = true;
    }
    carry = buf[nfull];
    __builtin_amdgcn_wave_barrier();
    c0 = endbits & 31;
    gw += nfull;
  }
  __device__ __forceinline__ void emit_run_u(u32 *buf, int lane, u32 bit, u32 count) {
    while (count) {
      const u32 m = count < 32 ? count : 32;
      emit_u(buf, lane, bit ? (m == 32 ? 0xFFFFFFFFu : ((1u << m) - 1)) : 0u, m);
      count -= m;
    }
  }
  // pack one round: rH = hi before the shift, rK = k | u << 8 per lane (0 for lanes without a symbol)
  __device__ __forceinline__ void pack(u32 *buf, int lane, u32 slow_threshold, u32 rH, u32 rK) {
    const u32 k = rK & 0xFF, u = rK >> 8;
    const bool flag = k != 0;
    // pending underflow before each symbol: segmented running sum of u, restarted by every emitting symbol
    const u32 S = wave_inclusive_sum(u);
    const u64 fm = __ballot(flag);
    const u64 upto = fm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1));  // flags at lanes <= lane
    const int f = upto ? 63 - __clzll((long long)upto) : -1;             // last emitting lane <= lane
    const u32 sbase = __shfl(S - u, f < 0 ? 0 : f, 64);                   // sum before that lane
    const u32 U = f < 0 ? pend + S : S - sbase;                           // pending after this lane
    const u32 P = __builtin_amdgcn_update_dpp(pend, U, 0x138, 0xF, 0xF, false);  // wave_shr:1, lane 0 keeps pend
    const u32 top = flag ? (k == 32 ? rH : (rH >> (32 - k))) : 0u;        // the k agreed bits
    const u32 msb = flag ? (top >> (k - 1)) : 0u;
    const u32 rest = (k > 1) ? (top & ((1u << (k - 1)) - 1)) : 0u;
    const u32 pend_out = __builtin_amdgcn_readlane(U, 63);
    if (__any(flag && P > slow_threshold)) {
      // an underflow run longer than 32 bits (about once per 2^32 symbols): walk the round serially
      for (int j = 0; j < 64; j++) {
        const u32 kj = __builtin_amdgcn_readlane(k, j);
        if (!kj) continue;
        const u32 Pj = __builtin_amdgcn_readlane(P, j), mj = __builtin_amdgcn_readlane(msb, j);
        const u32 rj = __builtin_amdgcn_readlane(rest, j);
        emit_u(buf, lane, mj, 1);
        emit_run_u(buf, lane, mj ^ 1, Pj);
        if (kj > 1) emit_u(buf, lane, rj, kj - 1);
      }
      pend = pend_out;
      return;
    }
    const u32 nbits = flag ? k + P : 0u;
    const u32 incl = wave_inclusive_sum(nbits);
    const u32 total = __builtin_amdgcn_readlane(incl, 63);
    const u32 o = c0 + incl - nbits;
    for (int w = lane; w < AC_ROUND_WORDS; w += 64) buf[w] = (w == 0) ? carry : 0u;
    __builtin_amdgcn_wave_barrier();
    if (flag) {
      const u32 run = (msb || P == 0) ? 0u : (P == 32 ? 0xFFFFFFFFu : ((1u << P) - 1));
      const u64 v = ((u64)msb << (P + k - 1)) | ((u64)run << (k - 1)) | rest;
      if (nbits > 32) {
        lds_place(buf, (u32)(v >> 32), nbits - 32, o);
        lds_place(buf, (u32)v, 32, o + nbits - 32);
      } else {
        lds_place(buf, (u32)v, nbits, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const u32 endbits = c0 + total;
    const u32 nfull = endbits >> 5;
    for (u32 w = lane; w < nfull; w += 64) {
      if (gw + w < wcap) dst[gw + w] = __builtin_bswap32(buf[w]);
      else over = true;
    }
    carry = buf[nfull];
    __builtin_amdgcn_wave_barrier();
    c0 = endbits & 31;
    gw += nfull;
    pend = pend_out;
  }
  // flush, arithmetic.cpp:160-169: bit 30 of lo, then pend+1 inverted copies, zero padding to a byte.  Returns bytes.
  __device__ __forceinline__ u32 finish(u32 *buf, int lane, u32 final_lo) {
    const u32 b30 = (final_lo >> 30) & 1;
    emit_u(buf, lane, b30, 1);
    emit_run_u(buf, lane, b30 ^ 1, pend + 1);
    const u64 bits = (u64)gw * 32 + c0;
    if (c0) {
      if (gw < wcap) { if (lane == 0) dst[gw] = __builtin_bswap32(carry); }
      else over = true;
    }
    return (u32)((bits + 7) >> 3);
  }
};

// Two blocks' rounds packed side by side (helper waves of ac_encode_rows_k serve two blocks each).  The same
// computation as AcSink::pack, written phase by phase for both blocks: a pack is a chain of LDS round trips (shuffle,
// zero fill, OR-ing the bits in, reading the finished words back), and one after the other the two chains added up to
// more than the helper's share of a SIMD allows once another shard's front stages run beside the coder (the chain wave
// then waited at the barrier for 27 % of its time).  Interleaved, one block's latency hides behind the other's work.
__device__ __forceinline__ void ac_pack2(AcSink *sk, u32 *buf0, u32 *buf1, int lane, u32 slow_threshold, const u32 *rH,
                                         const u32 *rK) {
  u32 *bufs[2] = {buf0, buf1};
  u32 k[2], P[2], msb[2], rest[2], pend_out[2], nbits[2], o[2], total[2];
  bool flag[2];
  bool slow = false;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    k[e] = rK[e] & 0xFF;
    const u32 u = rK[e] >> 8;
    flag[e] = k[e] != 0;
    const u32 S = wave_inclusive_sum(u);
    const u64 fm = __ballot(flag[e]);
    const u64 upto = fm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1));
    const int f = upto ? 63 - __clzll((long long)upto) : -1;
    const u32 sbase = __shfl(S - u, f < 0 ? 0 : f, 64);
    const u32 U = f < 0 ? sk[e].pend + S : S - sbase;
    P[e] = __builtin_amdgcn_update_dpp(sk[e].pend, U, 0x138, 0xF, 0xF, false);
    const u32 top = flag[e] ? (k[e] == 32 ? rH[e] : (rH[e] >> (32 - k[e]))) : 0u;
    msb[e] = flag[e] ? (top >> (k[e] - 1)) : 0u;
    rest[e] = (k[e] > 1) ? (top & ((1u << (k[e] - 1)) - 1)) : 0u;
    pend_out[e] = __builtin_amdgcn_readlane(U, 63);
    slow = slow || __any(flag[e] && P[e] > slow_threshold);
  }
  if (slow) {  // an underflow run longer than 32 bits (about once per 2^32 symbols): the serial walk, block by block
#pragma unroll
    for (int e = 0; e < 2; e++) {
      sk[e].to_plain(bufs[e], lane);
      sk[e].pack(bufs[e], lane, slow_threshold, rH[e], rK[e]);
      sk[e].to_buffered(bufs[e], lane);
    }
    return;
  }
  // The bits of a round are appended to what the buffer already holds and leave 64 words (one coalesced 256-byte
  // store) at a time, every 7-14 rounds: a store per round and block was ~24 bytes wide, and with stores in flight
  // behind every load the waits for the next operands had to drain them too.
#pragma unroll
  for (int e = 0; e < 2; e++) {
    nbits[e] = flag[e] ? k[e] + P[e] : 0u;
    const u32 incl = wave_inclusive_sum(nbits[e]);
    total[e] = __builtin_amdgcn_readlane(incl, 63);
    o[e] = sk[e].fill + incl - nbits[e];
    // zero the words this round reaches for the first time (the partial word at the end of the buffer keeps its bits)
    const u32 w0 = (sk[e].fill + 31) >> 5, w1 = (sk[e].fill + total[e] + 31) >> 5;
    for (u32 w = w0 + lane; w < w1; w += 64) bufs[e][w] = 0u;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < 2; e++) {
    if (flag[e]) {
      const u32 run = (msb[e] || P[e] == 0) ? 0u : (P[e] == 32 ? 0xFFFFFFFFu : ((1u << P[e]) - 1));
      const u64 v = ((u64)msb[e] << (P[e] + k[e] - 1)) | ((u64)run << (k[e] - 1)) | rest[e];
      if (nbits[e] > 32) {
        lds_place(bufs[e], (u32)(v >> 32), nbits[e] - 32, o[e]);
        lds_place(bufs[e], (u32)v, 32, o[e] + nbits[e] - 32);
      } else {
        lds_place(bufs[e], (u32)v, nbits[e], o[e]);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < 2; e++) {
    sk[e].fill += total[e];
    sk[e].pend = pend_out[e];
    while (sk[e].fill >= 2048) {  // 64 full words: out, and the rest moves to the front
      if (sk[e].gw + lane < sk[e].wcap) sk[e].dst[sk[e].gw + lane] = __builtin_bswap32(bufs[e][lane]);
      else sk[e].over = true;
      const u32 rem = (sk[e].fill - 2048 + 31) >> 5;  // words behind the 64 (at most 130)
      u32 t[3];
#pragma unroll
      for (int j = 0; j < 3; j++) t[j] = (lane + 64 * j < rem) ? bufs[e][64 + lane + 64 * j] : 0u;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (lane + 64 * j < rem) bufs[e][lane + 64 * j] = t[j];
      __builtin_amdgcn_wave_barrier();
      sk[e].fill -= 2048;
      sk[e].gw += 64;
    }
  }
}

template <bool GENERAL>
__global__ __launch_bounds__(128) void ac_encode_k(AcEncArgs a) {
  // chain -> helper, per symbol: {range received, lo + B} as latched by the systolic path (the helper redoes the symbol
  // from them) or, from the general path, the outcome itself {hi before the shift, k | u << 8}
  __shared__ uint2 rec[2][64];
  __shared__ u32 recfmt[2];      // 1: the round holds latched states
  __shared__ uint4 opsb[2][64];  // helper -> chain: operands of a round (slot = round & 1)
  __shared__ u32 oflag[2];       // ... and whether the round may take the systolic path
  __shared__ u32 buf[AC_BUF_WORDS];
  __shared__ u32 final_lo;

  const u32 blk = blockIdx.x;
  const u64 boff = (u64)blk * AC_BLOCK_SYMS;
  const u8 *s = a.sym + boff;
  const u32 n = (u32)((a.nsym - boff) < (u64)AC_BLOCK_SYMS ? (a.nsym - boff) : (u64)AC_BLOCK_SYMS);
  const int lane = lane_id();
  // Which of the two waves runs the chain?  The chain wave issues an instruction every slot its SIMD gives it, so
  // two chains on one SIMD both run at half speed -- and that is what happens to ~40 % of the blocks when two
  // launches (two shards in flight) are resident together, because the dispatcher places waves without knowing
  // their role.  Each wave therefore looks up how loaded its own SIMD already is (chain = 4, helper = 1, kept in
  // a device-wide table that every launch shares), and the wave on the lighter SIMD takes the chain.
  __shared__ u32 role_load[2];
  const u32 skey = simd_key();
  if (a.simd_load) {
    if (lane == 0) role_load[wave_id()] = atomicAdd(&a.simd_load[skey], 1u);
    __syncthreads();
  }
  const int chain_id = (a.simd_load && role_load[1] < role_load[0]) ? 1 : 0;
  const bool chain_wave = wave_id() == chain_id;
  if (a.simd_load && chain_wave && lane == 0) atomicAdd(&a.simd_load[skey], 3u);
  const u32 nrounds = (n + 63) >> 6;

  // the chain owns its SIMD's issue slots, whatever else lands there gets the gaps; the helper must not fall behind
  // the chain either when another shard's front stages fill the chip
  if (chain_wave) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(2);
  // ---- helper-wave state (bit sink) ----
  AcSink sink;
  sink.dst = (SCALCE_GLOBAL u32 *)(a.out + (u64)blk * a.out_stride);
  sink.wcap = a.out_cap / 4;
  // ---- chain-wave state ----
  u32 lo = 0, hi = 0xFFFFFFFFu;

  u64 prof_sys = 0, prof_rounds = 0;
  const u64 prof_t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
  // ---- operand pipeline (helper wave) ----
  // {g(lo), g(hi)} of the 64 symbols of round rr, lane l serving symbol 64 rr + l.  The context of lane l is the
  // two symbols before it: sy = symbols of round rr (one per lane), sy_prev = those of round rr - 1 for lanes 0, 1.
  auto sym_at = [&](u32 i) -> u32 { return i < n ? (u32)s[i] : 0u; };
  auto lookup = [&](u32 sy_prev, u32 sy, u32 base) -> uint4 {
    const u32 e63 = __builtin_amdgcn_readlane(sy_prev, 63), e62 = __builtin_amdgcn_readlane(sy_prev, 62);
    const u32 p1 = __builtin_amdgcn_update_dpp(e63, sy, 0x138, 0xF, 0xF, false);
    const u32 p0 = __builtin_amdgcn_update_dpp(e62, p1, 0x138, 0xF, 0xF, false);
    const u32 D1 = AC_D - 1;  // symbols >= AC_D raised E_SYMBOL at ingest; stay inside the table regardless
    const u32 c = sy < D1 ? sy : D1, q1 = p1 < D1 ? p1 : D1, q0 = p0 < D1 ? p0 : D1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (base + lane < n) v = a.tab[(q0 * AC_D + q1) * AC_D + c];
    return v;
  };
  // a round may take the systolic path when it is complete, not the first one (two raw symbols) and none of its
  // symbols is the last one of its context (c_hi == total: hi stays, ac_step)
  auto plain_ok = [&](const uint4 &v, u32 rr) -> u32 {
    return (rr > 0 && (rr << 6) + 64 <= n && !__any(v.w == 0xFFFFFFFFu)) ? 1u : 0u;
  };

  // outcome of a symbol from what its lane latched in the chain wave: hi before the shift, k agreed bits, u underflow
  // steps (the chain wave used to redo this itself after every round; the helper has the slack)
  auto outcome = [&](const uint2 v, const uint4 &o, u32 fmt) -> uint2 {
    if (!fmt) return v;
    const u32 inM = v.x, nlo = v.y;
    const u32 A = (u32)(((u64)inM * o.w + __umulhi(inM, o.z)) >> 32);
    const u32 B = (u32)(((u64)inM * o.y + __umulhi(inM, o.x)) >> 32);
    const u32 nhi = nlo + (A - B) - 1;
    const u32 k = ffbh_raw(nlo ^ nhi);
    const u32 c1 = ((~nlo | nhi) << 1) | 1u;
    const u32 u = ffbh_raw(c1 << k);
    // lo travels with bit 31 uncleared (sys_step): nlo = received + B, the stray bit is bit 31 of what was received
    return make_uint2(nhi ^ ((nlo - B) & 0x80000000u), k | (u << 8));
  };

  if (!chain_wave) {
    // ================= helper wave: operands two rounds ahead, bits one round behind =================
    sink.carry = ((u32)s[0] << 24) | ((n > 1 ? (u32)s[1] : 0u) << 16);  // raw first two symbols (:110-120)
    u32 sy_a, sy_b;  // symbols of rounds r + 1 and r + 2
    uint4 hist[3];   // operands of rounds r - 1, r, r + 1 (their LDS slots are long overwritten)
    {
      const u32 sy0 = sym_at(lane);
      sy_a = sym_at(64 + lane);
      sy_b = sym_at(128 + lane);
      const uint4 o0 = lookup(0u, sy0, 0), o1 = lookup(sy0, sy_a, 64);
      opsb[0][lane] = o0;
      opsb[1][lane] = o1;
      hist[0] = make_uint4(0, 0, 0, 0);
      hist[1] = o0;
      hist[2] = o1;
      const u32 ok1 = plain_ok(o1, 1);  // a wave-wide vote: not under the lane-0 branch
      if (lane == 0) { oflag[0] = 0; oflag[1] = ok1; }
    }
    __syncthreads();
    __syncthreads();  // the chain wave has read slot 0 (round 0 writes the operands of round 2 there)
    for (u32 r = 0; r < nrounds; r++) {
      const uint4 o2 = lookup(sy_a, sy_b, (r + 2) << 6);   // lands while the bits of round r - 1 are packed
      const u32 sy_c = sym_at(((r + 3) << 6) + lane);
      if (r > 0) {
        const uint2 v = outcome(rec[(r - 1) & 1][lane], hist[0], recfmt[(r - 1) & 1]);
        const bool valid = !(r == 1 && lane < 2);  // round 0: the two raw symbols carry no outcome
        sink.pack(buf, lane, a.slow_threshold, valid ? v.x : 0u, valid ? v.y : 0u);
      }
      opsb[r & 1][lane] = o2;                    // slot of round r: the chain wave took it a round ago
      const u32 ok2 = plain_ok(o2, r + 2);
      if (lane == 0) oflag[r & 1] = ok2;
      sy_a = sy_b;
      sy_b = sy_c;
      hist[0] = hist[1];
      hist[1] = hist[2];
      hist[2] = o2;
      __syncthreads();
    }
    {  // outcomes of the last round
      const u32 r = nrounds - 1;
      const u32 cnt = n - (r << 6);
      const uint2 v = outcome(rec[r & 1][lane], hist[0], recfmt[r & 1]);
      const bool valid = (u32)lane < cnt && !(r == 0 && lane < 2);
      sink.pack(buf, lane, a.slow_threshold, valid ? v.x : 0u, valid ? v.y : 0u);
    }
    const u32 bytes = sink.finish(buf, lane, final_lo);
    const bool over = __any(sink.over);  // (a block that ran out of room reports size 0: nothing frames bytes it does not hold)
    if (lane == 0) a.out_size[blk] = over ? 0u : bytes;
    if (over && lane == 0) dev_fail(a.err, E_ACOVERFLOW, blk, bytes);
  } else {
    // ================= chain wave: nothing but the coder state =================
    __syncthreads();  // operands of rounds 0 and 1 are in LDS
    uint4 cur = opsb[0][lane];
    __syncthreads();  // ... and slot 0 is in registers
    u32 cur_ok = 0;
    for (u32 r = 0; r < nrounds; r++) {
      const u32 base = r << 6;
      const uint4 ops = cur;
      const uint4 nxt = opsb[(r + 1) & 1][lane];  // needed at the end of the round
      const u32 nxt_ok = oflag[(r + 1) & 1];
      const u32 cnt = (n - base) < 64 ? (n - base) : 64;
      uint2 *rc = rec[r & 1];
      const u32 M0 = hi - lo + 1;  // 0 stands for 2^32 (full interval): only the general step can start from it
      bool done = false;
      if (!GENERAL && cur_ok && M0 != 0) {
        SysState st;
        const u64 pt0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
        u32 tlo, tM;
        sys_round(st, lo, M0, ops, tlo, tM);
        if (a.prof) { const u64 pt1 = __builtin_amdgcn_s_memtime(); prof_sys += pt1 - pt0; prof_rounds++; }
        // every lane hands the state it latched in its own step to the helper, which redoes the symbol from it
        const int q = lane & 3;
        const u32 inM = q == 0 ? st.kM[0] : q == 1 ? st.kM[1] : q == 2 ? st.kM[2] : st.kM[3];
        const u32 nlo = q == 0 ? st.nl[0] : q == 1 ? st.nl[1] : q == 2 ? st.nl[2] : st.nl[3];
        // Exit test.  The host only selects this path when no context total exceeds 2^29: every symbol then keeps an
        // interval of at least two values, so "all 32 bits agree" cannot happen; a range that renormalises to the full
        // 2^32 leaves M = 0 behind, which is absorbing in the plain step (D + 1 = 0 whatever the operands) and so is
        // still there in lane 63 at the end of the round.
        const u32 Mfin = __builtin_amdgcn_readlane(tM, 63);
        if (Mfin != 0 && !(a.test_poison && r % a.test_poison == 0)) {  // else (rare): redo the round below
          lo = __builtin_amdgcn_readlane(tlo, 63) & 0x7FFFFFFFu;
          hi = lo + Mfin - 1;
          rc[lane] = make_uint2(inM, nlo);
          if (lane == 0) recfmt[r & 1] = 1;
          done = true;
        }
      }
      if (!done) {  // general steps on lane 0: first round, last-of-context symbols, full interval, tails
        u32 glo = lo, ghi = hi;
        if (lane == 0) {
          recfmt[r & 1] = 0;
          for (u32 j = (r == 0) ? 2u : 0u; j < cnt; j++) {
            const uint4 g = make_uint4(__builtin_amdgcn_readlane(ops.x, j), __builtin_amdgcn_readlane(ops.y, j),
                                       __builtin_amdgcn_readlane(ops.z, j), __builtin_amdgcn_readlane(ops.w, j));
            u32 hbefore;
            const u32 ku = ac_step<GENERAL>(glo, ghi, g, hbefore);
            rc[j] = make_uint2(hbefore, ku);
          }
        }
        lo = __builtin_amdgcn_readfirstlane(glo);
        hi = __builtin_amdgcn_readfirstlane(ghi);
      }
      if (r + 1 == nrounds && lane == 0) final_lo = lo;
      cur = nxt;
      cur_ok = nxt_ok;
      __syncthreads();
    }
  }
  if (a.prof && chain_wave && lane == 0) {
    a.prof[blk * 3 + 0] = prof_sys;
    a.prof[blk * 3 + 1] = __builtin_amdgcn_s_memtime() - prof_t0;
    a.prof[blk * 3 + 2] = prof_rounds;
  }
  if (a.simd_load && lane == 0) atomicSub(&a.simd_load[skey], chain_wave ? 4u : 1u);
}

// ---- encoder, several blocks per workgroup -------------------------------------------------------
// Same coder, laid out for SIMD time instead of latency.  ac_encode_k spends a whole wavefront on one chain: 63 of
// 64 lanes execute garbage in every step.  Here groups of R lanes of the chain wave carry independent blocks
// (R = 16: four blocks, R = 8: eight): the state of a block walks the R lanes of its group while the same 14
// instructions + 1 wait state serve all groups.  A round is R symbols per block; the parallel recomputation runs
// after every round; operands, outcomes and the barrier with the helpers go by super-rounds of 64 symbols per block
// as in ac_encode_k.  Helper waves (two blocks each) gather operands and pack bits.
//   R = 16: DPP row_ror:1 in every step -- lane 0 of a row takes what lane 15 left, rounds follow each other without
//           any hand-over.  Per 64 symbols of four blocks the chain wave spends ~5000 cycles instead of 4 x 4400:
//           0.57 x the SIMD time per block at 1.13 x its latency.
//   R = 8:  no DPP mode rotates inside 8 lanes: step 0 of a round reads the state through row_ror:9 (lane 0 <- lane 7,
//           lane 8 <- lane 15: the last lane of the group, where the previous round left it), steps 1..7 through
//           row_shr:1.  ~0.35 x the SIMD time per block.
template <int R, int S>
__device__ __forceinline__ void sys_step_rows(SysState &st, u32 &tlo, u32 &tM, const uint4 &ops) {
  constexpr int Q = S & 3;
  // write enable: the quad of lane S of every group (R = 8: two groups per 16-lane row)
  constexpr int BM = R == 16 ? (1 << (S >> 2)) : ((1 << (S >> 2)) | (1 << ((S >> 2) + 2)));
  constexpr int CTRL = R == 16 ? 0x121 /* row_ror:1 */ : (S == 0 ? 0x129 /* row_ror:9 */ : 0x111 /* row_shr:1 */);
  st.kM[Q] = __builtin_amdgcn_update_dpp(st.kM[Q], tM, CTRL, 0xF, BM, false);
  const u32 M = st.kM[Q];
  const u32 A1 = (u32)(((u64)M * ops.w + (((u64)st.ones << 32) | __umulhi(M, ops.z))) >> 32);
  const u32 B = (u32)(((u64)M * ops.y + (((u64)st.zero << 32) | __umulhi(M, ops.x))) >> 32);
  const u32 D = A1 - B;
  if constexpr (R == 16)
    asm("v_add_u32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:%3" : "+v"(st.nl[Q]) : "v"(tlo), "v"(B), "n"(BM));
  else if constexpr (S == 0)
    asm("v_add_u32_dpp %0, %1, %2 row_ror:9 row_mask:0xf bank_mask:%3" : "+v"(st.nl[Q]) : "v"(tlo), "v"(B), "n"(BM));
  else
    asm("v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:%3" : "+v"(st.nl[Q]) : "v"(tlo), "v"(B), "n"(BM));
  __builtin_amdgcn_sched_barrier(0);
  const u32 nlo = st.nl[Q];
  const u32 t = renorm_count(nlo, D);
  tM = (D + 1) << t;
  tlo = nlo << t;
}
template <int R, int S, int E>
struct SysLoopRows {
  static __device__ __forceinline__ void run(SysState &st, u32 &tlo, u32 &tM, const uint4 &ops) {
    sys_step_rows<R, S>(st, tlo, tM, ops);
    SysLoopRows<R, S + 1, E>::run(st, tlo, tM, ops);
  }
};
template <int R, int E>
struct SysLoopRows<R, E, E> {
  static __device__ __forceinline__ void run(SysState &, u32 &, u32 &, const uint4 &) {}
};
// one round of R symbols for every group of the wave
template <int R>
__device__ __forceinline__ void sys_round_rows(SysState &st, u32 &tlo, u32 &tM, const uint4 &ops) {
  SysLoopRows<R, 0, R>::run(st, tlo, tM, ops);
}

template <bool GENERAL, int R>
__global__ __launch_bounds__(64 * (1 + 32 / R)) void ac_encode_rows_k(AcEncArgs a) {
  constexpr int NB = 64 / R;   // blocks per workgroup
  constexpr int NH = NB / 2;   // helper waves
  constexpr int NQ = 64 / R;   // rounds per super-round
  constexpr int AC4 = NB;
  // chain -> helpers, per symbol: {range received, lo + B} as latched by the systolic path (the helper redoes the symbol
  // from them: hi before the shift, k, u) or, from the general path, the outcome itself {hi before the shift, k | u << 8}
  __shared__ uint2 rec[2][AC4][64];
  __shared__ u32 recfmt[2][AC4];      // 1: that block's super-round holds latched states
  __shared__ uint4 opsb[2][AC4][64];  // helpers -> chain: operands of a super-round (slot = super-round & 1)
  __shared__ u32 oflag[2][AC4];       // ... and whether that block's super-round may take the systolic path
  __shared__ u32 bufs[NH][2][AC_BUF_WORDS];
  __shared__ u32 wave_simd[1 + NH];
  __shared__ u32 final_lo[AC4];

  const int lane = lane_id();
  // Roles.  The chain wave must have a SIMD to itself; the waves of a workgroup go to the CU's four SIMDs in turn, so
  // with five waves (R = 8) two of them share one.  The chain is the first wave whose SIMD no other wave of the
  // workgroup sits on, the others are helpers in wave order.
  if (lane == 0) wave_simd[wave_id()] = simd_key() & 3u;
  __syncthreads();
  int chain_w = 0;
  for (int w = NH; w >= 0; w--) {
    bool alone = true;
    for (int v = 0; v <= NH; v++) alone = alone && (v == w || wave_simd[v] != wave_simd[w]);
    if (alone) chain_w = w;
  }
  const int wv = wave_id() == chain_w ? 0 : (wave_id() < chain_w ? wave_id() + 1 : wave_id());  // 0 chain, 1.. helpers
  const u32 blk0 = blockIdx.x * AC4;
  auto block_len = [&](u32 b) -> u32 { return b < a.nblocks ? a.desc[b].n : 0u; };
  // super-rounds of the workgroup = those of its longest block (the last block of a stream may be short)
  u32 nmax = 0;
  for (int i = 0; i < AC4; i++) { const u32 x = block_len(blk0 + i); nmax = x > nmax ? x : nmax; }
  const u32 nsr = (nmax + 63) >> 6;

  if (wv != 0) {
    // ================= helper waves: two blocks each =================
    // beside other shards' front stages it is the helpers, not the chain, that fall behind (SCALCE_AC_PROF: the chain
    // waves then wait at the barrier for 27 % of their time, 0.9 % alone)
    switch (a.helper_prio) {
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 3: __builtin_amdgcn_s_setprio(3); break;
      default: break;
    }
    const int h = wv - 1;
    u32 *buf0 = bufs[h][0], *buf1 = bufs[h][1];
    AcSink sink[2];
    u32 wcap_all[2];
    bool in_place[2];
    const SCALCE_GLOBAL u8 *sp[2];        // (global, not generic: see SCALCE_GLOBAL)
    const SCALCE_GLOBAL u32x4 *tabp[2];
    u32 nb[2];
    // Operands and symbols in flight, in two register sets that alternate by name (the loop below runs two super-rounds
    // per iteration): what is requested at the top of one super-round is first touched in the next one, and the
    // barrier between them waits for LDS only.  Before, every super-round ended with the loads it had just issued
    // (copied at its end; __syncthreads() drains vmcnt anyway), so beside other shards' front stages -- when a fetch
    // takes a few microseconds -- the chain wave stood at the barrier for 18-25 % of its time.
    uint4 oA[2], oB[2];   // operands of super-round r + 2 (in use) / r + 3 (being fetched)
    u32 sA[2], sB[2];     // symbols of super-round r + 3 (in use) / r + 4 (being fetched)
    u32 e62[2], e63[2];   // the two symbols in front of the next super-round to be addressed
    auto sym_at = [&](int e, u32 i) -> u32 {  // unconditional load (a load behind a branch cannot be counted by s_waitcnt)
      const u32 v = sp[e][i < nb[e] ? i : 0u];
      return i < nb[e] ? v : 0u;
    };
    auto lookup = [&](int e, u32 sy, u32 base) -> uint4 {
      const u32 p1 = __builtin_amdgcn_update_dpp(e63[e], sy, 0x138, 0xF, 0xF, false);
      const u32 p0 = __builtin_amdgcn_update_dpp(e62[e], p1, 0x138, 0xF, 0xF, false);
      e63[e] = __builtin_amdgcn_readlane(sy, 63);
      e62[e] = __builtin_amdgcn_readlane(sy, 62);
      const u32 D1 = AC_D - 1;
      const u32 c = sy < D1 ? sy : D1, q1 = p1 < D1 ? p1 : D1, q0 = p0 < D1 ? p0 : D1;
      const u32x4 t = tabp[e][(q0 * AC_D + q1) * AC_D + c];
      const bool live = base + lane < nb[e];
      return make_uint4(live ? t.x : 0u, live ? t.y : 0u, live ? t.z : 0u, live ? t.w : 0u);
    };
    // outcome of a symbol from what its lane latched in the chain wave (the "parallel recomputation" of ac_encode_k,
    // done here where there is slack): hi before the shift, k agreed bits, u underflow steps
    auto outcome = [&](const uint2 v, const uint4 &o, u32 fmt) -> uint2 {
      if (!fmt) return v;
      const u32 inM = v.x, nlo = v.y;
      const u32 A = (u32)(((u64)inM * o.w + __umulhi(inM, o.z)) >> 32);
      const u32 B = (u32)(((u64)inM * o.y + __umulhi(inM, o.x)) >> 32);
      const u32 nhi = nlo + (A - B) - 1;
      const u32 k = ffbh_raw(nlo ^ nhi);
      const u32 c1 = ((~nlo | nhi) << 1) | 1u;
      const u32 u = ffbh_raw(c1 << k);
      // lo travels with bit 31 uncleared (sys_step): nlo = received + B, the stray bit is bit 31 of what was received
      return make_uint2(nhi ^ ((nlo - B) & 0x80000000u), k | (u << 8));
    };
    uint4 hist[2][3];  // operands of super-rounds r - 1, r, r + 1 (the chain wave has long overwritten their LDS slots)
    auto plain_ok = [&](int e, const uint4 &v, u32 rr) -> u32 {
      return (rr > 0 && (rr << 6) + 64 <= nb[e] && !__any(v.w == 0xFFFFFFFFu)) ? 1u : 0u;
    };
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const u32 b = blk0 + 2 * h + e;
      nb[e] = block_len(b);
      const AcBlockDesc dsc = nb[e] ? a.desc[b] : a.desc[0];
      sp[e] = (const SCALCE_GLOBAL u8 *)dsc.sym;
      tabp[e] = (const SCALCE_GLOBAL u32x4 *)dsc.tab;
      sink[e].dst = (SCALCE_GLOBAL u32 *)dsc.dst;
      wcap_all[e] = dsc.cap / 4;
      in_place[e] = (dsc.flags & AC_BLOCK_IN_PLACE) != 0;
      // In place: the words of super-round r's stores must lie in front of what this wave has taken out of the block for good.
      // The symbols of super-round r + 3 are in registers when super-round r packs (they were requested a super-round earlier
      // and looked up at its top), those of r + 4 are in flight: 16 words per super-round, two super-rounds of margin.
      sink[e].wcap = in_place[e] ? (wcap_all[e] < 32u ? wcap_all[e] : 32u) : wcap_all[e];
      if (nb[e]) sink[e].carry = ((u32)sp[e][0] << 24) | ((nb[e] > 1 ? (u32)sp[e][1] : 0u) << 16);
      e62[e] = e63[e] = 0;
      const uint4 o0 = lookup(e, sym_at(e, lane), 0), o1 = lookup(e, sym_at(e, 64 + lane), 64);
      oA[e] = lookup(e, sym_at(e, 128 + lane), 128);
      sA[e] = sym_at(e, 192 + lane);
      opsb[0][2 * h + e][lane] = o0;
      opsb[1][2 * h + e][lane] = o1;
      hist[e][0] = make_uint4(0, 0, 0, 0);
      hist[e][1] = o0;
      hist[e][2] = o1;
      const u32 ok1 = plain_ok(e, o1, 1);
      if (lane == 0) { oflag[0][2 * h + e] = 0; oflag[1][2 * h + e] = ok1; }
      sink[e].to_buffered(e ? buf1 : buf0, lane);  // the two raw symbols of the block open its buffer
    }
    __syncthreads();
    // ... and the chain wave has taken slot 0 into registers before super-round 0 puts the operands of super-round 2
    // there.  (Every later slot is read a whole super-round before it is overwritten; this first hand-over was not, and
    // once the helpers stopped waiting for their loads they could win the race: a block coded wrongly now and then.)
    __syncthreads();
    u64 hprof_wait = 0;
    const u64 hprof_t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    auto super_round = [&](auto first, const u32 r, uint4 (&o_use)[2], u32 (&s_use)[2], uint4 (&o_load)[2], u32 (&s_load)[2]) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        o_load[e] = lookup(e, s_use[e], (r + 3) << 6);
        s_load[e] = sym_at(e, ((r + 4) << 6) + lane);
      }
      if constexpr (!decltype(first)::value) {  // (super-round 0 has nothing to pack: peeled, so that every pass of the
                                                //  loop issues its stores -- see ac_pack2)
        u32 rH[2], rK[2];
#pragma unroll
        for (int e = 0; e < 2; e++)
          if (in_place[e]) { const u32 lim = (16u * (r + 2u)) >> a.inplace_shift; sink[e].wcap = lim < wcap_all[e] ? lim : wcap_all[e]; }
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const uint2 v = outcome(rec[(r - 1) & 1][2 * h + e][lane], hist[e][0], recfmt[(r - 1) & 1][2 * h + e]);
          const bool valid = (((r - 1) << 6) + lane < nb[e]) && !(r == 1 && lane < 2);
          rH[e] = valid ? v.x : 0u;
          rK[e] = valid ? v.y : 0u;
        }
        ac_pack2(sink, buf0, buf1, lane, a.slow_threshold, rH, rK);
      }
#pragma unroll
      for (int e = 0; e < 2; e++) {
        opsb[r & 1][2 * h + e][lane] = o_use[e];
        const u32 ok2 = plain_ok(e, o_use[e], r + 2);
        if (lane == 0) oflag[r & 1][2 * h + e] = ok2;
        hist[e][0] = hist[e][1];
        hist[e][1] = hist[e][2];
        hist[e][2] = o_use[e];
      }
      if (a.prof && h == 0) {
        const u64 w0 = __builtin_amdgcn_s_memtime();
        barrier_lds_only();
        hprof_wait += __builtin_amdgcn_s_memtime() - w0;
      } else {
        barrier_lds_only();
      }
    };
    if (nsr) super_round(std::true_type{}, 0u, oA, sA, oB, sB);
    for (u32 r = 1; r < nsr; r += 2) {
      super_round(std::false_type{}, r, oB, sB, oA, sA);
      if (r + 1 < nsr) super_round(std::false_type{}, r + 1, oA, sA, oB, sB);
    }
    if (a.prof && h == 0 && lane == 0) {
      a.prof[blockIdx.x * 5 + 3] = hprof_wait;
      a.prof[blockIdx.x * 5 + 4] = __builtin_amdgcn_s_memtime() - hprof_t0;
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      if (!nb[e]) continue;
      const u32 r = nsr - 1;
      const uint2 v = outcome(rec[r & 1][2 * h + e][lane], hist[e][0], recfmt[r & 1][2 * h + e]);
      const bool valid = ((r << 6) + lane < nb[e]) && !(r == 0 && lane < 2);
      u32 *buf = e ? buf1 : buf0;
      sink[e].wcap = wcap_all[e];  // (every symbol of the block has been taken: what is left may use all of it)
      sink[e].to_plain(buf, lane);
      sink[e].pack(buf, lane, a.slow_threshold, valid ? v.x : 0u, valid ? v.y : 0u);
      const u32 bytes = sink[e].finish(buf, lane, final_lo[2 * h + e]);
      const AcBlockDesc dsc = a.desc[blk0 + 2 * h + e];
      const bool over = __any(sink[e].over);
      if (lane == 0) *dsc.out_size = over ? 0u : bytes;
      if (over && lane == 0) dev_fail(dsc.err, E_ACOVERFLOW, dsc.index, bytes);
    }
  } else {
    // ================= chain wave: NB coder states, one per group of R lanes =================
    switch (a.chain_prio) {  // an immediate in the instruction
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    const int row = lane / R, col = lane % R;
    const u32 n_row = block_len(blk0 + row);
    const u32 nsr_row = (n_row + 63) >> 6;
    SysState st;
    asm("v_mov_b32 %0, -1" : "=v"(st.ones));
    asm("v_mov_b32 %0, 0" : "=v"(st.zero));
    st.kM[0] = st.kM[1] = st.kM[2] = st.kM[3] = 0;
    st.nl[0] = st.nl[1] = st.nl[2] = st.nl[3] = 0;
    // the travelling state; between rounds the one that matters sits in the last lane of the group.  M = 0 stands for 2^32.
    u32 tlo = 0, tM = 0;
    __syncthreads();  // operands of super-rounds 0 and 1 are in LDS
    uint4 cur[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) cur[q] = opsb[0][row][q * R + col];
    __syncthreads();  // slot 0 is in registers: the helpers may overwrite it (see their side)
    u32 cur_ok = 0;
    u64 prof_wait = 0;
    const u64 prof_t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    for (u32 r = 0; r < nsr; r++) {
      uint4 ops[NQ], nxt[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) { ops[q] = cur[q]; nxt[q] = opsb[(r + 1) & 1][row][q * R + col]; }
      const u32 nxt_ok = oflag[(r + 1) & 1][row];
      uint2 *rc = rec[r & 1][row];
      const u32 s_lo = tlo, s_M = tM;  // state at the start of the super-round (last lane of each group)
      // which rows may run the systolic path: flagged complete + plain by the helper, and not the full interval
      const u64 m15 = __ballot(col == R - 1 && s_M != 0);
      const bool row_plain = !GENERAL && cur_ok && ((m15 >> (row * R + R - 1)) & 1);
      const bool row_live = r < nsr_row;
      if (!GENERAL && __any(row_plain)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          sys_round_rows<R>(st, tlo, tM, ops[q]);
          // every lane hands the state it latched in its own step to the helper, which redoes the symbol from it
          const int sq = col & 3;
          const u32 inM = sq == 0 ? st.kM[0] : sq == 1 ? st.kM[1] : sq == 2 ? st.kM[2] : st.kM[3];
          const u32 nlo = sq == 0 ? st.nl[0] : sq == 1 ? st.nl[1] : sq == 2 ? st.nl[2] : st.nl[3];
          rc[q * R + col] = make_uint2(inM, nlo);
        }
      }
      // Exit test.  The host only selects this path when no context total exceeds 2^29: every symbol then keeps an
      // interval of at least two values (range > 2^30), so "all 32 bits agree" cannot happen.  The other exit, a range
      // that renormalises to the full 2^32, leaves M = 0 behind -- and M = 0 is absorbing in the plain step
      // (D + 1 = 0 whatever the operands), so it is still there in the last lane at the end of the super-round.
      if (a.test_poison && r % a.test_poison == 0) tM = 0;  // tests: drive the redo path
      const u64 badm = __ballot(col == R - 1 && tM == 0);
      const bool row_bad = ((badm >> (row * R + R - 1)) & 1) != 0;
      const bool need_general = row_live && (!row_plain || row_bad);
      if (col == R - 1) recfmt[r & 1][row] = need_general ? 0u : 1u;
      const u64 gm = __ballot(need_general);
      if (gm) {  // rare: first super-round, tails, a step that needs the general path -- row by row on lane 0
        for (int rw = 0; rw < AC4; rw++) {
          if (!((gm >> (rw * R)) & 1)) continue;
          const u32 nr = block_len(blk0 + rw);
          const u32 rest = nr - (r << 6);
          const u32 jend = rest < 64 ? rest : 64, jstart = (r == 0) ? 2u : 0u;
          // The systolic path lets lo travel with a stray bit 31 (sys_step), which is removed here.  Under GENERAL the
          // state only ever comes from ac_step<true>, where bit 31 of lo is real (lo = 1.., hi = 0.. is a state the
          // reference's 32-bit coder runs into once a context total passes 2^30): lo goes through untouched and
          // hi = lo + M - 1 modulo 2^32 is exact for inverted intervals as well.
          u32 glo = __builtin_amdgcn_readlane(s_lo, rw * R + R - 1);
          if (!GENERAL) glo &= 0x7FFFFFFFu;
          u32 ghi = glo + __builtin_amdgcn_readlane(s_M, rw * R + R - 1) - 1;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            for (u32 c = 0; c < (u32)R; c++) {
              const u32 j = q * R + c;
              const uint4 g = make_uint4(__builtin_amdgcn_readlane(ops[q].x, rw * R + c), __builtin_amdgcn_readlane(ops[q].y, rw * R + c),
                                         __builtin_amdgcn_readlane(ops[q].z, rw * R + c), __builtin_amdgcn_readlane(ops[q].w, rw * R + c));
              if (j >= jstart && j < jend && lane == 0) {
                u32 hbefore;
                const u32 ku = ac_step<GENERAL>(glo, ghi, g, hbefore);
                rec[r & 1][rw][j] = make_uint2(hbefore, ku);
              }
            }
          }
          glo = __builtin_amdgcn_readfirstlane(glo);
          ghi = __builtin_amdgcn_readfirstlane(ghi);
          if (lane == rw * R + R - 1) { tlo = glo; tM = ghi - glo + 1; }
        }
      }
      if (col == R - 1 && r + 1 == nsr_row) final_lo[row] = tlo & 0x7FFFFFFFu;
#pragma unroll
      for (int q = 0; q < NQ; q++) cur[q] = nxt[q];
      cur_ok = nxt_ok;
      if (a.prof) {
        const u64 w0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        prof_wait += __builtin_amdgcn_s_memtime() - w0;
      } else {
        __syncthreads();
      }
    }
    if (a.prof && lane == 0) {
      a.prof[blockIdx.x * 5 + 0] = prof_wait;
      a.prof[blockIdx.x * 5 + 1] = __builtin_amdgcn_s_memtime() - prof_t0;
      a.prof[blockIdx.x * 5 + 2] = nsr;
    }
  }
}

// [u32 size][bytes] framing (arithmetic.cpp:335-336,355-356): block b goes to dst_off[b]
struct AcFrameLen {
  const u32 *sizes;
  __device__ u64 operator()(u64 b) const { return 4ull + sizes[b]; }
};
__global__ __launch_bounds__(256) void ac_frame_k(const u8 *blocks, u64 stride, const u32 *sizes, const u64 *dst_off,
                                                 u8 *out) {
  // Block b's bytes go behind its size word at out + dst_off[b] + 4: any alignment.  Output words are produced aligned to
  // 4 bytes from two neighbouring source words (funnel shift); the few bytes in front of the first aligned word, the size
  // word and the tail are written byte by byte.  (One byte per store, as this kernel first was, ran at 1 TB/s.)
  const u32 b = blockIdx.y;
  const u32 sz = sizes[b];
  u8 *dst = out + dst_off[b];
  const u32 *src = reinterpret_cast<const u32 *>(blocks + (u64)b * stride);  // 16-byte aligned (stride is a multiple of 16)
  u8 *data = dst + 4;
  const u32 head = (u32)((4 - ((u64)data & 3)) & 3);       // data bytes in front of the first aligned output word
  const u32 lead = head < sz ? head : sz;
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    for (int k = 0; k < 4; k++) dst[k] = (u8)(sz >> (8 * k));
    const u8 *sb = reinterpret_cast<const u8 *>(src);
    for (u32 i = 0; i < lead; i++) data[i] = sb[i];
  }
  const u32 nwords = (sz - lead) >> 2;                      // aligned output words, word w holds source bytes lead + 4 w ..
  const u32 sh = (lead & 3) * 8;
  u32 *dw = reinterpret_cast<u32 *>(data + lead);
  for (u32 w0 = (u32)(t * 4); w0 < nwords; w0 += gridDim.x * blockDim.x * 4) {
    const u32 j = (lead >> 2) + w0;                         // source word that holds byte lead + 4 w0 (lead < 4: j = w0)
    u32 sv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) sv[k] = src[j + k];         // the block buffer is padded beyond its 10 MiB
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (w0 + k < nwords) dw[w0 + k] = sh ? ((sv[k] >> sh) | (sv[k + 1] << (32 - sh))) : sv[k];
  }
  if (t == 1) {                                             // the tail behind the last aligned word
    const u8 *sb = reinterpret_cast<const u8 *>(src);
    for (u32 i = lead + 4 * nwords; i < sz; i++) data[i] = sb[i];
  }
}

// The same framing for a WINDOW [lo, hi) of the framed stream, written to out[0 .. hi - lo): what a caller takes out slice by
// slice without the stream ever existing as a whole on the device (scalce_batch_qual_window; `out` may be pinned host
// memory, the stores then go over the link in runs of 16 bytes per thread).  Block b's frame is the virtual byte string
// V_b = [size, 4 bytes LE][its sz coded bytes] at stream offset dst_off[b]; the part of it inside the window goes out.
// `out` is 4-byte aligned; output words are produced aligned to `out`.
__global__ __launch_bounds__(256) void ac_frame_window_k(const u8 *blocks, u64 stride, const u32 *sizes, const u64 *dst_off,
                                                        u64 lo, u64 hi, u8 *out, u32 first_block) {
  const u32 b = first_block + blockIdx.y;                  // (the host knows which blocks the window touches)
  const u32 sz = sizes[b];
  const u64 f0 = dst_off[b], f1 = f0 + 4ull + sz;          // the frame's place in the stream
  const u64 a = f0 > lo ? f0 : lo, e = f1 < hi ? f1 : hi;  // its part inside the window
  if (a >= e) return;
  const u8 *sb = blocks + (u64)b * stride;                 // 16-byte aligned (stride is a multiple of 16)
  const u32 *src = reinterpret_cast<const u32 *>(sb);
  auto vbyte = [&](u64 p) -> u8 {                          // byte of the stream at offset p, f0 <= p < f1
    const u64 r = p - f0;
    return r < 4 ? (u8)(sz >> (8 * r)) : sb[r - 4];
  };
  // words: stream offsets p with (p - lo) % 4 == 0, wholly inside [max(a, f0 + 4), e)
  const u64 d0 = a > f0 + 4 ? a : f0 + 4;                  // first data byte inside the window
  u64 w_lo = lo + (((d0 - lo) + 3) & ~3ull);               // first aligned position at or behind it
  if (w_lo > e) w_lo = e;
  const u64 nwords = (e - w_lo) >> 2;
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0)
    for (u64 p = a; p < w_lo; p++) out[p - lo] = vbyte(p);
  if (t == 1)
    for (u64 p = w_lo + 4 * nwords; p < e; p++) out[p - lo] = vbyte(p);
  const u64 s0 = w_lo - f0 - 4;                            // source byte of the first word (>= 0 whenever nwords > 0)
  const u32 sh = (u32)(s0 & 3) * 8;
  u32 *dw = reinterpret_cast<u32 *>(out + (w_lo - lo));
  for (u64 w0 = t * 4; w0 < nwords; w0 += (u64)gridDim.x * blockDim.x * 4) {
    const u64 j = (s0 >> 2) + w0;
    u32 sv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) sv[k] = src[j + k];        // the block buffer is padded beyond its 10 MiB
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (w0 + k < nwords) dw[w0 + k] = sh ? ((sv[k] >> sh) | (sv[k + 1] << (32 - sh))) : sv[k];
  }
}

// follows the [u32 size][bytes] chain of a framed stream: out[2i] = byte offset of block i's data, out[2i+1] = its size;
// out[2 nblk] != 0 when the stream ends before `nblk` frames do
__global__ void ac_frame_walk_k(const u8 *in, u64 nbytes, u32 nblk, u64 *out) {
  if (threadIdx.x || blockIdx.x) return;
  u64 pos = 0;
  u64 bad = 0;
  for (u32 i = 0; i < nblk; i++) {
    if (bad || pos + 4 > nbytes) { bad = 1; out[2 * (u64)i] = 0; out[2 * (u64)i + 1] = 0; continue; }
    const u32 sz = (u32)in[pos] | ((u32)in[pos + 1] << 8) | ((u32)in[pos + 2] << 16) | ((u32)in[pos + 3] << 24);
    out[2 * (u64)i] = pos + 4;
    out[2 * (u64)i + 1] = sz;
    pos += 4 + (u64)sz;
    if (pos > nbytes) bad = 1;
  }
  out[2 * (u64)nblk] = bad;
}

// ---- decoder (next row): ac_decoder::read_single, arithmetic.cpp:196-244 -------------------
// One wavefront per block.  The reference divides by the current range to get `count` and then searches the
// symbol linearly (:199-209).  Here the 64 lanes evaluate the encoder's own interval bounds of all 80 symbols
// of the context at once -- floor(range * cum[s+1] / total) through the same reciprocal fractions -- and one
// ballot finds the symbol whose interval holds the code value: for v = code - lo,
//   count >= cum[s]  <=>  v >= floor(range * cum[s] / total),   count < cum[s+1]  <=>  v < floor(range * cum[s+1] / total),
// so the search is equivalent and no division is left.  The table row of the context is the one dependent
// memory access per symbol (the context is only known once the previous symbol is out).
// Bit reader of one block, one wavefront: the 64 lanes hold the next 64 big-endian words of the stream (one coalesced
// load per 2048 bits, the following window already in flight), a word is handed out by v_readlane.  (Read byte by
// byte at the point of use, each refill cost four dependent memory round trips: a third of the decoder's time.)
struct AcBitReader {
  const u8 *in;
  u32 insz;      // bytes of the block; zeros are read past its end
  u32 base;      // byte position of lane 0's word of the current window
  u32 wcur, wnext;
  u32 wi;        // next word of the current window (wave-uniform)
  u64 win;       // the next `wb` bits, left aligned
  u32 wb;
  __device__ __forceinline__ u32 load_window(u32 pos0, int lane) const {
    const u32 pos = pos0 + 4u * (u32)lane;
    u32 w = 0;
    if (pos + 4 <= insz) {
      u32 raw;
      __builtin_memcpy(&raw, in + pos, 4);  // any alignment
      w = __builtin_bswap32(raw);
    } else {
      for (int k = 0; k < 4; k++) w = (w << 8) | (pos + k < insz ? (u32)in[pos + k] : 0u);
    }
    return w;
  }
  __device__ __forceinline__ void start(const u8 *p, u32 n, u32 pos0, int lane) {
    in = p; insz = n; base = pos0;
    wcur = load_window(pos0, lane);
    wnext = load_window(pos0 + 256, lane);
    wi = 0; win = 0; wb = 0;
    refill(lane);
    refill(lane);
  }
  __device__ __forceinline__ void refill(int lane) {  // wb <= 32: append one word
    const u32 w = __builtin_amdgcn_readlane(wcur, wi);
    win |= (u64)w << (32 - wb);
    wb += 32;
    if (__builtin_expect(++wi == 64, 0)) {
      wi = 0;
      base += 256;
      wcur = wnext;
      wnext = load_window(base + 256, lane);
    }
  }
  __device__ __forceinline__ u32 get(u32 cnt, int lane) {  // cnt in 1..32
    const u32 r = (u32)(win >> (64 - cnt));
    win <<= cnt;
    wb -= cnt;
    if (wb <= 32) refill(lane);
    return r;
  }
  __device__ __forceinline__ u32 get0(u32 cnt, int lane) {  // cnt in 0..32
    const u32 r = (u32)((win >> 1) >> (63 - cnt));
    win <<= cnt;
    wb -= cnt;
    if (__builtin_expect(wb <= 32, 0)) refill(lane);
    return r;
  }
};
// Renormalisation of the decoder (arithmetic.cpp:225-239) without its loop and without a branch in the usual case:
// k agreed leading bits leave, then u underflow steps; lo, hi and the code register move by k + u bits at once.
__device__ __forceinline__ void ac_dec_renorm(u32 &lo, u32 &hi, u32 &code, u32 nlo, u32 nhi, AcBitReader &br, int lane) {
  const u32 x = nlo ^ nhi;
  const u32 k = x ? (u32)__clz(x) : 32u;
  const u32 l1 = (u32)((u64)nlo << k);
  const u32 h1 = (u32)(((u64)nhi << k) | ((1ull << k) - 1));
  const u32 y = (l1 & ~h1) << 1;
  const u32 u = (u32)__clz(~y);  // positions from bit 30 down where lo has 1 and hi has 0; bit 0 of ~y is set
  const u32 t = k + u;
  const u32 top = u ? 0x80000000u : 0u;
  lo = (u32)((u64)l1 << u) & ~top;
  hi = (u32)(((u64)h1 << u) | ((1ull << u) - 1)) | top;
  if (__builtin_expect(t <= 32, 1)) {
    const u32 bits = br.get0(t, lane);
    code = ((u32)((u64)code << t) | bits) ^ top;  // u steps of code = ((code ^ 0x40000000) << 1) | bit flip what ends up in bit 31
  } else {  // more than 32 bits at once: rare
    code = (u32)((u64)code << k) | br.get(k, lane);
    code = ((code << u) ^ 0x80000000u) | br.get(u, lane);
  }
}
struct AcDecArgs {
  const u8 *in;          // framed stream
  const u64 *blk_off;    // byte offset of each block's payload (after its size word)
  const u32 *blk_size;
  u64 nsym;
  const uint4 *tab;      // [6400][80] {g(lo), g(hi)} as built by ac_table_k
  u8 *out;
};
__global__ __launch_bounds__(64) void ac_decode_k(AcDecArgs a) {
  const u32 blk = blockIdx.x;
  const u64 boff = (u64)blk * AC_BLOCK_SYMS;
  const u32 n = (u32)((a.nsym - boff) < (u64)AC_BLOCK_SYMS ? (a.nsym - boff) : (u64)AC_BLOCK_SYMS);
  const u8 *in = a.in + a.blk_off[blk];
  const u32 insz = a.blk_size[blk];
  u8 *out = a.out + boff;
  const int lane = lane_id();
  u32 p0 = insz > 0 ? in[0] : 0, p1 = insz > 1 ? in[1] : 0;
  if (lane == 0) { out[0] = (u8)p0; if (n > 1) out[1] = (u8)p1; }
  if (p0 >= AC_D) p0 = AC_D - 1;  // corrupt stream: stay inside the tables
  if (p1 >= AC_D) p1 = AC_D - 1;
  AcBitReader br;
  br.start(in, insz, 2, lane);
  auto getbits = [&](u32 cnt) -> u32 { return br.get(cnt, lane); };
  u32 lo = 0, hi = 0xFFFFFFFFu, code = getbits(32);
  // lane j keeps symbol (i & 63) == j until 64 are gathered; the two raw symbols sit in lanes 0 and 1
  u32 outacc = lane == 0 ? (insz > 0 ? (u32)in[0] : 0u) : (lane == 1 ? (insz > 1 ? (u32)in[1] : 0u) : 0u);
  for (u32 i = 2; i < n; i++) {
    const uint4 *row = a.tab + (u64)(p0 * AC_D + p1) * AC_D;
    const uint4 e0 = row[lane];
    const uint4 e1 = lane < 16 ? row[64 + lane] : make_uint4(0, 0, 0, 0);
    const u32 R = hi - lo, v = code - lo;
    // upper bound of symbol `lane` (and 64 + lane) relative to lo; the last symbol of a context reaches the range itself
    const bool last0 = e0.w == 0xFFFFFFFFu, last1 = e1.w == 0xFFFFFFFFu;
    const u32 U0 = mulfrac(R, e0.z, e0.w), U1 = mulfrac(R, e1.z, e1.w);
    const u64 m0 = __ballot(last0 || v < U0);
    const u64 m1 = __ballot(lane < 16 && (last1 || v < U1));
    u32 sidx, A, B;
    bool is_last;
    if (m0) {
      sidx = (u32)__ffsll((long long)m0) - 1;
      A = __builtin_amdgcn_readlane(U0, sidx);
      is_last = __builtin_amdgcn_readlane((u32)last0, sidx) != 0;
      B = sidx ? __builtin_amdgcn_readlane(U0, sidx - 1) : 0u;
    } else {
      const u32 t = m1 ? (u32)__ffsll((long long)m1) - 1 : 15u;  // corrupt stream: last symbol
      sidx = 64 + t;
      A = __builtin_amdgcn_readlane(U1, t);
      is_last = __builtin_amdgcn_readlane((u32)last1, t) != 0;
      B = t ? __builtin_amdgcn_readlane(U1, t - 1) : __builtin_amdgcn_readlane(U0, 63);
    }
    ac_dec_renorm(lo, hi, code, lo + B, is_last ? hi : lo + A - 1, br, lane);
    outacc = ((u32)lane == (i & 63)) ? sidx : outacc;
    if ((i & 63) == 63) out[(i & ~63u) + lane] = (u8)outacc;  // 64 symbols per store
    p0 = p1;
    p1 = sidx;
  }
  // tail of the last partial group
  const u32 done = n & ~63u;
  if ((n & 63) && (u32)lane < (n & 63)) out[done + lane] = (u8)outacc;
}

// ---- decoder with the hot contexts' rows in LDS ------------------------------------------------------------
// ac_decode_k's symbol costs one dependent 1280-byte row fetch from L2 (~0.4 us).  Two things shorten it:
//  * compact rows: the decoder only needs the UPPER bound g(hi) of every symbol (the lower bound is the neighbour's),
//    and only for the symbols that occur -- `rows[ctx]` holds S1 <= 64 entries of 8 bytes for the symbols
//    smin - 1 .. smax (entry 0 is the bound below the span).  A code value outside the span (a symbol whose scaled count
//    is 1 everywhere: possible, rare) goes through the full row exactly as ac_decode_k does;
//  * the rows of the W x W most frequent contexts (the W symbols with the largest totals) sit in LDS (128 KB, two
//    blocks per workgroup share them), addressed by the symbols' ranks, no map in between;
// and the next row is requested as soon as the symbol is known, before the renormalisation and the bit refill.
constexpr u32 AC_DEC_CACHE_ENTRIES = 19968;  // x 8 bytes = 156 KB of the CU's 160 KB of LDS (round 4: 128 KB; a row outside costs a symbol 110 ns more)
struct AcDecCachedArgs {
  AcDecArgs d;
  const uint2 *rows;   // [6400][S1]
  u32 smin, S1, W, nblk;
  u8 hot[32];          // the W cached symbols, by rank
  u8 rank[80];         // rank of a symbol, 0xFF = not cached
};
__global__ __launch_bounds__(256) void ac_dec_rows_k(const uint4 *tab, u32 smin, u32 S1, uint2 *rows) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6400u * S1) return;
  const u32 ctx = i / S1, j = i % S1;
  uint2 v = make_uint2(0, 0);  // smin = 0: nothing lies below the span, the bound is 0
  if (smin + j >= 1) { const uint4 e = tab[(u64)ctx * AC_D + (smin + j - 1)]; v = make_uint2(e.z, e.w); }
  rows[i] = v;
}



// ---- decoder, the loop written for the scalar unit (round 4): ac_decode_lean_k in ~45 instructions per symbol --------
// ac_decode_lean_k's loop is ~80 instructions as the compiler emits it (230 ns per symbol: a lone wavefront issues one
// instruction in four to five cycles, and the chain of a block is nothing but issue slots): a dozen s_cselect / s_cmp to
// materialise conditions, renorm_count on the vector unit and back (v_mov, v_bfe, s_nop, v_readfirstlane), two scalar
// multiplications for the address of the next row, v_cmp + v_cndmask for the output lane, eleven register moves where the
// two paths of the loop meet.  Here:
//   * ONE exit test per symbol: s_ff1 of the ballot returns -1 for "no lane" -- j <= 0 covers a code value below or above the
//     compact row, and the full range too (M = 0 makes every U zero, so no lane answers);
//   * the LDS offset of the next row is a sum of two per-lane constants read with v_readlane: lane l of a compact row IS
//     symbol smin - 1 + l, cW[l] = its rank * W * S1 * 8 (as the first symbol of a context), cS[l] = its rank * S1 * 8 (as
//     the second); a symbol outside the cache carries 2^30, so one comparison says "not in LDS";
//   * the renormalisation count entirely on the scalar unit (flbit, shift, and, add, flbit), and code - lo moves through
//     ONE 64-bit shift together with the next bits of the stream ({v - B : window} << t), which also covers t = 0;
//   * the decoded symbol goes to its output lane with v_writelane;
//   * the plain path is an inner loop of its own, the generic step (the reference's own registers, as in ac_decode_lean_k)
//     sits outside it: nothing is copied where they meet.
// Same arguments, launch shapes and conditions as ac_decode_lean_k (no context total above 2^29, symbol 79 outside the span).
#ifndef AC_DEC_TIGHT_CXX
#define AC_DEC_TIGHT_CXX 0  // 1: the plain path as the compiler schedules it (comparison; 178 ns per symbol)
#endif
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void ac_decode_tight_k(AcDecCachedArgs a) {
  __shared__ uint2 cache[AC_DEC_CACHE_ENTRIES + 64];
  const u32 W = a.W, S1 = a.S1;
  for (u32 i = threadIdx.x; i < W * W * S1; i += blockDim.x) {
    const u32 slot = i / S1, j = i % S1;
    const u32 ctx = (u32)a.hot[slot / W] * AC_D + a.hot[slot % W];
    cache[i] = a.rows[(u64)ctx * S1 + j];
  }
  for (u32 i = W * W * S1 + threadIdx.x; i < W * W * S1 + 64; i += blockDim.x) cache[i] = make_uint2(0, 0);
  __syncthreads();
  const u32 blk = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave_id());  // (uniform to the compiler as well: pointers and sizes in SGPRs)
  if (blk >= a.nblk) return;
  const u64 boff = (u64)blk * AC_BLOCK_SYMS;
  const u32 n = __builtin_amdgcn_readfirstlane((u32)((a.d.nsym - boff) < (u64)AC_BLOCK_SYMS ? (a.d.nsym - boff) : (u64)AC_BLOCK_SYMS));
  const u8 *in = a.d.in + a.d.blk_off[blk];
  const u32 insz = a.d.blk_size[blk];
  u8 *out = a.d.out + boff;
  const int lane = lane_id();
  u32 p0 = insz > 0 ? in[0] : 0, p1 = insz > 1 ? in[1] : 0;
  if (lane == 0) { out[0] = (u8)p0; if (n > 1) out[1] = (u8)p1; }
  if (p0 >= AC_D) p0 = AC_D - 1;
  if (p1 >= AC_D) p1 = AC_D - 1;
  p0 = __builtin_amdgcn_readfirstlane(p0);
  p1 = __builtin_amdgcn_readfirstlane(p1);
  constexpr u32 FAR = 1u << 30;  // "this symbol's rows are not in LDS"
  const u32 smin1 = a.smin - 1u;
  const u32 my_sym = smin1 + (u32)lane;  // (lane 0: the bound below the span, never decoded)
  const u32 my_rank = (lane >= 1 && (u32)lane < S1 && my_sym < AC_D) ? (u32)a.rank[my_sym] : 0xFFu;
  const u32 cS = my_rank < W ? my_rank * S1 * 8u : FAR;
  const u32 cW = my_rank < W ? my_rank * W * S1 * 8u : FAR;
  const u64 span = S1 >= 64 ? ~0ull : ((1ull << S1) - 1);
  const u32 lane8 = (u32)lane * 8u;
  auto off_of = [&](u32 sy, u32 scale) -> u32 {  // sy wave-uniform; only for the two raw symbols and after the generic step
    const u32 r = sy < AC_D ? (u32)a.rank[sy] : 0xFFu;
    return r < W ? r * scale : FAR;
  };
  typedef u32 u32x2 __attribute__((ext_vector_type(2)));
  const u32 cache_lds = (u32)(uintptr_t)(__attribute__((address_space(3))) u32x2 *)cache;
  const SCALCE_GLOBAL u32x2 *rows_g = (const SCALCE_GLOBAL u32x2 *)a.rows;
  // the row of context (c0, c1): from LDS at offset `off` when off < FAR, else from the row table (padded by 64 entries)
  auto fetch = [&](u32 c0, u32 c1, u32 off) -> uint2 {
    if (__builtin_expect(off < FAR, 1)) {
      const u32x2 x = *(__attribute__((address_space(3))) u32x2 *)(uintptr_t)(cache_lds + off + lane8);
      return make_uint2(x.x, x.y);
    }
    const u32x2 x = rows_g[(u64)(c0 * AC_D + c1) * S1 + lane];
    return make_uint2(x.x, x.y);
  };
  AcBitReader br;
  br.start(in, insz, 2, lane);
  // State as in ac_decode_lean_k: lo, M = hi - lo + 1 (0 stands for 2^32), v = code - lo.
  u32 lo = 0, M = 0, v = br.get(32, lane);
  u32 outacc = lane == 0 ? (insz > 0 ? (u32)in[0] : 0u) : (lane == 1 ? (insz > 1 ? (u32)in[1] : 0u) : 0u);
  u32 qW = off_of(p1, W * S1 * 8u);  // what the last symbol adds to the offset of the NEXT context's row
  uint2 e = fetch(p0, p1, off_of(p0, W * S1 * 8u) + off_of(p1, S1 * 8u));  // (a sum with FAR in it is >= FAR)
  u32 i = 2;
  while (i < n) {
    // ---- the plain path, symbol after symbol ----
    u32 rare = 0;
    if (!AC_DEC_TIGHT_CXX) {
      // The loop by hand: 42 instructions per symbol (the compiler's version of the same statements, below: 59 -- where its
      // paths meet it copies the state from register to register and turns conditions into masks and back).  Fixed
      // registers, so that the halves of a 64-bit pair can be named: s41 lo, s42 M, s57 v (the high half of the pair that
      // takes in the stream's bits), s44 qW, s46 / s47 / s[48:49] the bit reader's wb / wi / window, v[106:107] the row
      // entry of this lane; i = s40 + m0 (s40 a multiple of 64, m0 the output lane), s61 = i - bound counts up to 0 (the
      // carry of its increment ends the loop).  The two symbols in front are not carried along: they are in the output
      // lanes (outacc) for the rare paths that need them.  Leaves the loop: at i = n (reason 0), in front of a symbol the
      // plain step cannot take (reason 1, nothing of the state touched), behind the symbol whose refill used up the
      // reader's 64-word window (s61 is set to -1: reason 0 with wi = 64).
      // (every scalar through v_readfirstlane: to the compiler some of them depend on the lane -- they never do)
      auto rfl = [](u32 x) -> u32 { return (u32)__builtin_amdgcn_readfirstlane(x); };
      auto rfl64 = [&](u64 x) -> u64 { return ((u64)rfl((u32)(x >> 32)) << 32) | (u64)rfl((u32)x); };
      u32 win_lo = rfl((u32)br.win), win_hi = rfl((u32)(br.win >> 32));
      u32 zi = rfl(i), zlo = rfl(lo), zM = rfl(M), zv = rfl(v), zqW = rfl(qW), zwb = rfl(br.wb), zwi = rfl(br.wi);
      const u64 zout = rfl64((u64)(uintptr_t)out), zrows = rfl64((u64)(uintptr_t)rows_g), zspan = rfl64(span);
      const u32 zneg = rfl(i - n), zsmin1 = rfl(smin1), zS1 = rfl(S1);   // i - n: negative, i < n here
      asm volatile(
          "s_mov_b32 s68, m0\n\t"
          "s_and_b32 m0, %[i], 63\n\ts_andn2_b32 s40, %[i], 63\n\ts_mov_b32 s61, %[neg]\n\t"
          "s_mov_b32 s41, %[lo]\n\ts_mov_b32 s42, %[M]\n\ts_mov_b32 s57, %[v]\n\ts_mov_b32 s44, %[qW]\n\t"
          "s_mov_b32 s46, %[wb]\n\ts_mov_b32 s47, %[wi]\n\ts_mov_b32 s48, %[wlo]\n\ts_mov_b32 s49, %[whi]\n\t"
          "v_mov_b32 v106, %[ex]\n\tv_mov_b32 v107, %[ey]\n\tv_mov_b32 v101, 0\n\ts_mov_b32 %[rare], 0\n"
          "Ltop_%=:\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mul_hi_u32 v100, s42, v106\n\t"
          "v_mad_u64_u32 v[102:103], s[64:65], s42, v107, v[100:101]\n\t"
          "v_cmp_lt_u32_e32 vcc, s57, v103\n\t"
          "s_and_b64 s[62:63], vcc, %[span]\n\t"
          "s_ff1_i32_b64 s50, s[62:63]\n\t"
          // the next row is asked for before anything else is looked at -- the recurrence of a block is row -> U -> j -> row.
          // With j = -1 or 0 (the generic step's business) or a context outside the cache the address is nonsense: LDS
          // answers a read beyond its end with zeros, and both ways out fetch the row again before anybody reads it.
          "v_readlane_b32 s53, %[cS], s50\n\t"
          "s_add_u32 s53, s53, s44\n\t"
          "v_add_u32_e32 v104, s53, %[ldsb]\n\t"
          "ds_read_b64 v[106:107], v104\n\t"
          "s_cmp_lt_i32 s50, 1\n\t"
          "s_cbranch_scc1 Lgeneric_%=\n\t"
          // From here on the chain that sets the pace of a block -- A, B -> D -> count t -> new M and v -> next U -- with what
          // does not depend on it between its links.  An s_cmp and its s_cbranch stay together: every scalar ALU instruction
          // in between would overwrite SCC.
          "s_add_i32 s59, s50, -1\n\t"
          "v_readlane_b32 s51, v103, s50\n\t"           // A
          "v_readlane_b32 s52, v103, s59\n\t"           // B
          "v_readlane_b32 s44, %[cW], s50\n\t"          //   qW of the next symbol
          "s_sub_u32 s42, s51, s52\n\t"                 // A - B = D + 1: what the new M is made of
          "s_add_i32 s54, s50, %[smin1]\n\t"            //   the symbol
          "s_add_u32 s55, s42, -1\n\t"                  // D
          "s_add_u32 s41, s41, s52\n\t"                 //   nlo = lo + B
          "s_cmp_ge_u32 s53, 0x40000000\n\t"
          "s_cbranch_scc1 Lfar_%=\n"
          "Lhave_%=:\n\t"
          "s_flbit_i32_b32 s59, s55\n\t"                // c
          "v_writelane_b32 %[outacc], s54, m0\n\t"      //   the symbol into its output lane
          "s_lshr_b32 s59, 0x7fffffff, s59\n\t"
          "s_sub_u32 s57, s57, s52\n\t"                 //   v - B
          "s_and_b32 s59, s59, s41\n\t"
          "s_mov_b32 s56, s49\n\t"                      //   the stream's next 32 bits below it
          "s_add_u32 s59, s59, s55\n\t"
          "s_flbit_i32_b32 s58, s59\n\t"                // t
          "s_lshl_b32 s42, s42, s58\n\t"                // M = (D + 1) << t
          "s_lshl_b64 s[56:57], s[56:57], s58\n\t"      // v
          "s_lshl_b32 s41, s41, s58\n\t"
          "s_lshl_b64 s[48:49], s[48:49], s58\n\t"
          "s_sub_u32 s46, s46, s58\n\t"
          "s_cmp_le_u32 s46, 32\n\t"
          "s_cbranch_scc1 Lrefill_%=\n"
          "Lrefilled_%=:\n\t"
          "s_add_u32 m0, m0, 1\n\t"
          "s_cmp_eq_u32 m0, 64\n\t"
          "s_cbranch_scc1 Lstore_%=\n"
          "Lstored_%=:\n\t"
          "s_add_u32 s61, s61, 1\n\t"                   // carry: the bound is reached
          "s_cbranch_scc0 Ltop_%=\n\t"
          "s_branch Lout_%=\n"
          "Lrefill_%=:\n\t"                       // wb <= 32: the next word of the window behind the bits held
          "v_readlane_b32 s66, %[wcur], s47\n\t"
          "s_mov_b32 s67, 0\n\t"
          "s_sub_u32 s59, 32, s46\n\t"
          "s_lshl_b64 s[66:67], s[66:67], s59\n\t"
          "s_or_b64 s[48:49], s[48:49], s[66:67]\n\t"
          "s_add_u32 s46, s46, 32\n\t"
          "s_add_u32 s47, s47, 1\n\t"
          "s_cmp_eq_u32 s47, 64\n\t"
          "s_cbranch_scc0 Lrefilled_%=\n\t"
          "s_mov_b32 s61, -1\n\t"                // the window is used up: this symbol is the last of the run
          "s_branch Lrefilled_%=\n"
          "Lstore_%=:\n\t"                        // 64 symbols, one per lane
          "v_or_b32_e32 v104, s40, %[lane]\n\t"
          "global_store_byte v104, %[outacc], %[out]\n\t"
          "s_add_u32 s40, s40, 64\n\t"
          "s_mov_b32 m0, 0\n\t"
          "s_branch Lstored_%=\n"
          "Lfar_%=:\n\t"                          // the next context's row is not in LDS: from the row table
          "s_add_i32 s63, m0, -1\n\t"             // (its first symbol is the one decoded before this one: the lane in front)
          "s_and_b32 s63, s63, 63\n\t"
          "s_nop 3\n\t"
          "v_readlane_b32 s63, %[outacc], s63\n\t"
          "s_min_u32 s63, s63, 0x4f\n\t"
          "s_mul_i32 s63, s63, 0x50\n\t"
          "s_add_u32 s63, s63, s54\n\t"
          "s_mul_i32 s63, s63, %[S1]\n\t"
          "v_add_u32_e32 v104, s63, %[lane]\n\t"
          "v_lshlrev_b32_e32 v104, 3, v104\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"                 // (the read that went beyond LDS has answered: nothing else writes the pair)
          "global_load_dwordx2 v[106:107], v104, %[rows]\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "s_branch Lhave_%=\n"
          "Lgeneric_%=:\n\t"
          "s_mov_b32 %[rare], 1\n"
          "Lout_%=:\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "s_add_u32 %[i], s40, m0\n\t"
          "s_mov_b32 m0, s68\n\t"
          "s_mov_b32 %[lo], s41\n\ts_mov_b32 %[M], s42\n\ts_mov_b32 %[v], s57\n\ts_mov_b32 %[qW], s44\n\t"
          "s_mov_b32 %[wb], s46\n\ts_mov_b32 %[wi], s47\n\t"
          "s_mov_b32 %[wlo], s48\n\ts_mov_b32 %[whi], s49\n\t"
          "v_mov_b32 %[ex], v106\n\tv_mov_b32 %[ey], v107\n"
          : [i] "+s"(zi), [lo] "+s"(zlo), [M] "+s"(zM), [v] "+s"(zv), [qW] "+s"(zqW), [wb] "+s"(zwb),
            [wi] "+s"(zwi), [wlo] "+s"(win_lo), [whi] "+s"(win_hi), [ex] "+v"(e.x), [ey] "+v"(e.y), [outacc] "+v"(outacc),
            [rare] "=&s"(rare)
          : [neg] "s"(zneg), [span] "s"(zspan), [smin1] "s"(zsmin1), [S1] "s"(zS1), [out] "s"(zout), [rows] "s"(zrows), [cS] "v"(cS),
            [cW] "v"(cW), [ldsb] "v"(cache_lds + lane8), [wcur] "v"(br.wcur), [lane] "v"((u32)lane)
          : "memory", "vcc", "scc", "s68", "s40", "s41", "s42", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
            "s54", "s55", "s56", "s57", "s58", "s59", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "v100", "v101", "v102",
            "v103", "v104", "v106", "v107");
      i = zi; lo = zlo; M = zM; v = zv; qW = zqW; br.wb = zwb; br.wi = zwi;
      br.win = ((u64)win_hi << 32) | win_lo;
      if (br.wi == 64) {  // (AcBitReader::refill's other half: the next 64 words of the block)
        br.wi = 0;
        br.base += 256;
        br.wcur = br.wnext;
        br.wnext = br.load_window(br.base + 256, lane);
      }
      // the two symbols in front of symbol i (the generic step's context): its output lanes
      p0 = (u32)__builtin_amdgcn_readlane(outacc, (i - 2u) & 63u);
      p1 = (u32)__builtin_amdgcn_readlane(outacc, (i - 1u) & 63u);
      p0 = p0 < AC_D ? p0 : AC_D - 1u;
      p1 = p1 < AC_D ? p1 : AC_D - 1u;
      if (!rare) continue;
    }
    while (AC_DEC_TIGHT_CXX && i < n) {
      // U[l] = floor(M * g[l] / 2^64): what symbol l's upper bound adds to lo
      const u32 U = (u32)(((u64)M * e.y + __umulhi(M, e.x)) >> 32);
      const u64 m = __ballot(v < U) & span;
      int j;
      asm("s_ff1_i32_b64 %0, %1" : "=s"(j) : "s"(m));  // first symbol whose upper bound lies above the code value; -1: none
      if (__builtin_expect(j <= 0, 0)) { rare = 1; break; }
      const u32 A = (u32)__builtin_amdgcn_readlane(U, j);
      const u32 B = (u32)__builtin_amdgcn_readlane(U, j - 1);
      const u32 off = qW + (u32)__builtin_amdgcn_readlane(cS, j);
      qW = (u32)__builtin_amdgcn_readlane(cW, j);
      const u32 sidx = smin1 + (u32)j;
      const uint2 e_next = fetch(p1, sidx, off);  // the next context is known: ask for its row before anything else
      p0 = p1;
      p1 = sidx;
      const u32 nlo = lo + B, D = A - 1u - B;
      const u32 c = (u32)__builtin_clz(D);                       // D >= 1: every symbol keeps two values or more
      const u32 t = (u32)__builtin_clz((nlo & (0x7FFFFFFFu >> c)) + D);   // renorm_count, on the scalar unit
      M = (D + 1u) << t;
      lo = nlo << t;
      // {v - B : the next 32 bits of the stream} << t: the high word is the new v (t = 0 .. 31 bits come in)
      v = (u32)(((((u64)(v - B)) << 32) | (u64)(u32)(br.win >> 32)) << t >> 32);
      br.win <<= t;
      br.wb -= t;
      if (__builtin_expect(br.wb <= 32, 0)) br.refill(lane);
      asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(outacc) : "s"(sidx), "s"(i & 63u));  // (M0 holds nothing in this kernel)
      if (__builtin_expect((i & 63u) == 63u, 0)) out[(i & ~63u) + lane] = (u8)outacc;
      e = e_next;
      i++;
    }
    if (!rare) break;
    // ---- below or above the symbols the compact rows hold, the last symbol of a context, the full range: the full row,
    // on the reference's own registers (lo without the stray bit, hi, code), exactly as ac_decode_lean_k ----
    {
      if (M == 0u) lo = 0u;
      else if ((u32)(lo + M - 1u) < lo) lo ^= 0x80000000u;
      u32 hi = lo + M - 1u, code = lo + v;
      const u32 R = hi - lo, vv = code - lo;
      bool is_last = false;
      u32 A, B, sidx;
      const uint4 *row = a.d.tab + (u64)(p0 * AC_D + p1) * AC_D;
      const uint4 f0 = row[lane];
      const uint4 f1 = lane < 16 ? row[64 + lane] : make_uint4(0, 0, 0, 0);
      const bool last0 = f0.w == 0xFFFFFFFFu, last1 = f1.w == 0xFFFFFFFFu;
      const u32 U0 = mulfrac(R, f0.z, f0.w), U1 = mulfrac(R, f1.z, f1.w);
      const u64 m0 = __ballot(last0 || vv < U0);
      const u64 m1 = __ballot(lane < 16 && (last1 || vv < U1));
      if (m0) {
        sidx = (u32)__ffsll((long long)m0) - 1;
        A = (u32)__builtin_amdgcn_readlane(U0, sidx);
        is_last = __builtin_amdgcn_readlane((u32)last0, sidx) != 0;
        B = sidx ? (u32)__builtin_amdgcn_readlane(U0, sidx - 1) : 0u;
      } else {
        const u32 t = m1 ? (u32)__ffsll((long long)m1) - 1 : 15u;  // corrupt stream: last symbol
        sidx = 64 + t;
        A = (u32)__builtin_amdgcn_readlane(U1, t);
        is_last = __builtin_amdgcn_readlane((u32)last1, t) != 0;
        B = t ? (u32)__builtin_amdgcn_readlane(U1, t - 1) : (u32)__builtin_amdgcn_readlane(U0, 63);
      }
      p0 = p1;
      p1 = sidx;
      qW = off_of(p1, W * S1 * 8u);
      e = fetch(p0, p1, off_of(p0, W * S1 * 8u) + off_of(p1, S1 * 8u));
      ac_dec_renorm(lo, hi, code, lo + B, is_last ? hi : lo + A - 1, br, lane);
      M = hi - lo + 1u;
      v = code - lo;
      outacc = ((u32)lane == (i & 63)) ? sidx : outacc;
      if ((i & 63) == 63) out[(i & ~63u) + lane] = (u8)outacc;
      i++;
    }
  }
  const u32 done = n & ~63u;
  if ((n & 63) && (u32)lane < (n & 63)) out[done + lane] = (u8)outacc;
}

// ---- self-test: closed-form step vs the reference's literal loop, on random and crafted states ------
__device__ __forceinline__ u32 mix32(u32 x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void ac_selftest_k(u64 n, u32 seed, int general, u32 *mismatch /*[0]=count,[1..5]=first case*/) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 r0 = mix32((u32)i * 2654435761u + seed), r1 = mix32(r0 + 0x9E3779B9u), r2 = mix32(r1 + 0x85EBCA6Bu),
      r3 = mix32(r2 + 0xC2B2AE35u), r4 = mix32(r3 + 0x27D4EB2Fu);
  u32 lo = r0, hi = r1;
  if (lo > hi) { const u32 t = lo; lo = hi; hi = t; }
  u32 d = (r2 >> (r4 & 15)) | 2u;                 // totals of every magnitude, >= 2
  if (d >= 0xFFFFFFFEu) d = 0xFFFFFFFDu;
  u32 c_lo = r3 % d, c_hi = c_lo + 1 + (r4 >> 8) % (d - c_lo);
  if (general != 1) {
    // production path: well-formed states only -- lo = 0.., hi = 1.., not (lo = 01.. and hi = 10..), total <= 2^30
    lo &= 0x7FFFFFFFu; hi |= 0x80000000u;
    if ((lo & 0x40000000u) && !(hi & 0x40000000u)) lo &= 0x3FFFFFFFu;
    d = (d & 0x3FFFFFFFu) | 2u;
    if ((i & 15) == 9) d = 0x40000000u;            // the largest total the fast path accepts
    c_lo = r3 % d; c_hi = c_lo + 1 + (r4 >> 8) % (d - c_lo);
  }
  switch (i & 7) {
    case 1: lo = 0; hi = 0xFFFFFFFFu; break;       // block start / after a 32-bit agreement: R + 1 wraps
    case 2: c_hi = d; break;                       // last symbol of the context
    case 3: if (general == 1) hi = lo + (r4 & 7);       // tiny ranges: inverted intervals, new lo == new hi
            else { c_hi = c_lo + 1; if ((r4 & 3) == 0) { lo = 0x3FFFFFFFu - (r3 & 0xFFFF); hi = 0xC0000000u + (r2 & 0xFFFF); } } break;
    case 4: lo = 0x7FFFFFF0u + (r3 & 15); hi = 0x80000000u + (r4 & 0xFFFF);
            if (general != 1) { lo = 0x3FFFFFF0u + (r3 & 15); hi = 0xC0000000u + (r4 & 0xFFFF); } break;  // near the midpoint
    case 5: c_lo = 0; break;
    case 6:  // the new interval is an aligned power-of-two block: renormalises to the full 2^32 without x == 0
      lo = 0x40000000u; hi = 0xBFFFFFFFu; d = 4u << (r3 & 7); c_lo = d / 4; c_hi = d / 2; break;
    default: break;
  }
  // literal reference step (arithmetic.cpp:122-152)
  u32 rlo = lo, rhi = hi, rk = 0, ru = 0, rhb;
  {
    const u64 range = (u64)(u32)(hi - lo) + 1;
    rhi = (u32)(lo + (range * c_hi) / d - 1);
    rlo = (u32)(lo + (range * c_lo) / d);
    rhb = rhi;
    for (;;) {
      if ((rhi & 0x80000000u) == (rlo & 0x80000000u)) rk++;
      else if (!(rhi & 0x40000000u) && (rlo & 0x40000000u)) { ru++; rlo &= 0x3FFFFFFFu; rhi |= 0x40000000u; }
      else break;
      rlo <<= 1;
      rhi = (rhi << 1) | 1;
      if (rk + ru > 70) break;
    }
  }
  // closed form through the reciprocal table entries
  const u64 glo = c_lo ? recip_frac(c_lo, d) : 1ull;
  const u64 ghi = (c_hi == d) ? ~0ull : recip_frac(c_hi, d);
  u32 flo = lo, fhi = hi, fhb;
  const uint4 gg = make_uint4((u32)glo, (u32)(glo >> 32), (u32)ghi, (u32)(ghi >> 32));
  u32 ku;
  if (general == 2) {
    // the plain step exactly as the encoder uses it: (lo, M) state, fall back to the general step on a 0 return
    u32 M = fhi - flo + 1, kup = 0;
    bool done = false;
    if (M != 0 && c_hi != d) {
      // ac_step_plain's exit test reads lane 0 of the wave; give every lane its own verdict here
      u32 plo = flo, pM = M, phb;
      const u32 A = 0, B = 0;
      (void)A; (void)B;
      const u32 e = ac_step_plain(plo, pM, gg, phb, kup);
      if (e != 0) { flo = plo; fhi = plo + pM - 1; fhb = phb; done = true; }
    }
    ku = done ? kup : ac_step<false>(flo, fhi, gg, fhb);
  } else {
    ku = general ? ac_step<true>(flo, fhi, gg, fhb) : ac_step<false>(flo, fhi, gg, fhb);
  }
  const bool ordered = rk == 0 || true;
  // in the literal loop the agreeing-bit phase and the underflow phase can only interleave as k then u
  const bool ok = ordered && flo == rlo && fhi == rhi && (ku & 0xFF) == rk && (ku >> 8) == ru && fhb == rhb;
  if (!ok && atomicAdd(&mismatch[0], 1u) == 0) {
    mismatch[1] = lo; mismatch[2] = hi; mismatch[3] = c_lo; mismatch[4] = c_hi; mismatch[5] = d;
  }
}

}  // namespace scalce
