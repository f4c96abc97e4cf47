// Weights-resident DATAFLOW backward through time for the per-rank shape of BASELINE configs[2] (b = 128 rows per GPU, 4 x LSTM(1024), bf16):
// autograd's backward of nn.LSTM (models.py:156,164) as ONE persistent launch instead of T + 3 wavefront launches + T + 3 element-wise launches.
//
// Per cell (l, t), t = T-1 .. 0:   dh = dG^l_{t+1} . W_hh^l  +  dG^{l+1}_t . W_ih^{l+1}  (+ dy_t for the top layer),   K = 2 x 4H = 8192,
// then the gate-derivative math gives dG^l_t [128 x 4H] and the carried dc.  The contraction runs over GATE ROWS, so a workgroup that owned
// whole hidden units would have to take in all 2 MB of [dG^l_{t+1} | dG^{l+1}_t] every step (4 x the forward pass's intake).  Instead:
//   * 256 workgroups, 64 per layer; workgroup (l, blk, kq) keeps the weights of 64 hidden units (blk) x ONE GATE's K-quarter (kq: columns
//     1024 kq .. + 1023 of both W_hh^T and W_ih^T) in registers: 64 x 2048 bf16 = 256 KB, as MFMA A-fragments (wave w: k in [256 w, +256) of the
//     quarter), and takes in only ITS quarter of the two dG blocks per step: 128 rows x 2 x 1024 k = 512 KB (the forward pass's intake), by
//     LDS-DMA into per-wave rings, exactly the forward kernel's operand path (rnn_persist.hip);
//     Slot order of a step: ALL x slots (the layer above is a diagonal ahead) before the h slots, so that two microseconds of work that
//     does not depend on the layer's own flags stand between a publish and the next poll (both 64-row halves' accumulators are live);
//   * the four K-slice partials of the waves are exchanged through LDS (as in the forward pass), which leaves the workgroup with its
//     K-QUARTER partial of dh [128 rows x 64 units]; the four workgroups (blk, 0..3) exchange those through a 2-deep ring in global memory:
//     workgroup kq receives BATCH ROWS 32 kq .. + 31 of all 64 units from all four, sums them in the fixed order kq = 0, 1, 2, 3, and does
//     the gate-derivative math for 32 rows x 64 units, one row x 8 consecutive units per lane (dc carried in registers for the whole pass):
//     a row's 64 units are 128 contiguous bytes per gate, so every saved-state load and dG store moves whole memory lines (a 16-unit x
//     128-row split would move 32-byte pieces: 3 x the issue time and 2.4 x the latency per instruction, tests/tuning/vmem).  The payload
//     is its own flag: every fp32 word carries the phase of its ring slot's use in its lowest mantissa bit (one ulp of a K-quarter partial
//     sum that is rounded to bf16 a moment later), so the producer neither drains nor raises anything and the consumer polls the data itself;
//   * dG^l_t goes out write-through, drained, and the workgroup raises dflags[l][t][j]; consumers (the same layer's 64 workgroups for step
//     t - 1, the layer below's 64 for step t -- every one of them reads ONE gate's columns of ALL units) poll the 64 words of (l, t).
// Every dG element is written once per launch; the exchange ring slot of step t is rewritten at step t - 2, when every reader has long
// published dflags[l][t - 1] (which the writer waited for).  Spins are bounded and end in a status record, as in the forward kernel.
// The top layer's output gradient comes as dy [T][B][H] fp32 (added in front of the gate math).
#include "common.hpp"
#include "tile_pipe.hpp"
#include "kernels.hpp"
#include "persist_common.hpp"

namespace {

constexpr int PH = 1024, PNL = 4, PB = 128;
constexpr int LDG = 4 * PH + 64;                  // leading dimension of the dG buffers (elements): fixed, every offset is a literal
constexpr int PWG = 256;
constexpr int RS = 6, SLOT = 4096, RING = RS * SLOT;
constexpr int SCR_OFF = 4 * RING, SCR_BYTES = 16 * 4096;      // reduction scratch: region (destination wave d, source wave s) at (4 d + s) x 4 KB
constexpr int PLDS = SCR_OFF + SCR_BYTES;         // = 160 KB, all of the CU's LDS
constexpr size_t EXCH_BYTES = (size_t)PNL * 2 * 16 * 4 * 4 * 8192;      // [layer][parity of t][blk][dst kq][src kq][plane 2][row 32][8 unit groups] float4 = 16 MB

struct PersistBwdArgs {
  int T;
  int Btot, row0;
  const void* w_hhT[PNL]; const void* w_ihT[PNL]; long ldw_hhT[PNL]; long ldw_ihT[PNL];      // [H][ldw] bf16: row = hidden unit / input feature, column = gate row
  const float* dy;                 // [T][Btot][H] fp32: gradient w.r.t. the top layer's output
  const void* gates[PNL];          // [T][Btot][4H] bf16 (saved by the forward pass)
  const void* cs[PNL];             // [T][Btot][H] bf16
  void* dG[PNL];                   // [T][Btot][LDG] bf16 (out)
  float* exch;                     // EXCH_BYTES
  uint32_t* dflags;                // [PNL][T][64]: the dG^l_t tile of workgroup j is in memory
  uint32_t* status;
  float* poison;                   // optional: receives a quiet NaN when the launch gives up (mvae_rnn_bwd_desc.poison)
  uint32_t spin_limit;
  int safe;
  unsigned long long* dbg;
};

#ifdef MVAE_TUNING
#define PBWD_STAMP(K) do { if (p.dbg && tid == 0) p.dbg[(((long)layer * T + t) * 64 + j) * 8 + (K)] = wall_clock64(); } while (0)
#else
#define PBWD_STAMP(K) do { } while (0)
#endif

typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;

// 16-byte write-through buffer store.  Observed on gfx950 (tests/tuning/persist/ab_persist_bwd.py, first row-split build): with the scalar
// offset in an SGPR the compiler emits no wait state between a buffer_store_dwordx4 and a VALU instruction that overwrites its data
// registers (the documented exemption for SGPR offsets), and dword 1 of the stored vector then carried the NEXT value written to that
// register in lanes 12-15 of every 16.  So: the whole offset goes through the VGPR (literal soffset 0: the compiler's hazard rule applies)
// and a few idle cycles follow the store.
__device__ __forceinline__ void store_b128_wt(u32x4_t v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 16);
  asm volatile("s_nop 3" ::: "memory");
}

// Poll the 64 flag words at byte offset `off` of the [status | flags] block: one 256-byte sc1 load per round, addressed through a buffer
// resource (an SGPR base + lane * 4: no 64-bit per-lane pointer to keep alive across the step).  Returns false on timeout / abort.
__device__ __forceinline__ bool wait_flags_b(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t lane4, uint32_t limit) {
  for (uint32_t it = 0; it < limit; ++it) {
    const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r, lane4, off, 16);
    asm volatile("" ::: "memory");                 // re-issued every round
    if (__builtin_amdgcn_ballot_w64(v != 0u) == ~0ull) return true;
    if ((it & 63) == 63) {
      const uint32_t st = __builtin_amdgcn_raw_buffer_load_b32(r, 0u, 0u, 16);
      asm volatile("" ::: "memory");
      if (__builtin_amdgcn_readfirstlane(st) != 0u) return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

// HAS_X: layers < top (second K-segment = the layer above's dG of the same step); the top layer adds dy instead.
template <bool HAS_X>
__device__ __forceinline__ void persist_bwd_body(const PersistBwdArgs& p, char* smem, int layer, int blk, int kq) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
  const int n = lane & 15, q = lane >> 4;
  const int j = 4 * blk + kq;                        // this workgroup's flag word

  // ---- resident weights: A fragments, lane (m = n, kg = q), m-tile g = units 64 blk + 16 g + m, k = 1024 kq + 256 w + 32 kb + 8 q .. + 8
  uint4 wx[4][8], wh[4][8];
  {
    const bf16_t* wr = reinterpret_cast<const bf16_t*>(p.w_hhT[layer]);
    const bf16_t* wi = reinterpret_cast<const bf16_t*>(HAS_X ? p.w_ihT[layer + 1] : p.w_hhT[layer]);
    const long ldr = p.ldw_hhT[layer], ldi = HAS_X ? p.ldw_ihT[layer + 1] : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const long row = 64 * blk + 16 * g + n;
        const long col = 1024 * kq + 256 * wave + 32 * kb + 8 * q;
        if (HAS_X) wx[g][kb] = *reinterpret_cast<const uint4*>(wi + row * ldi + col);
        wh[g][kb] = *reinterpret_cast<const uint4*>(wr + row * ldr + col);
      }
  }
  // The MFMA / LDS-exchange lane (n, q) of wave w holds, per half hf, batch row 16 (4 hf + w) + n, units 16 g + 4 q .. + 3 of the four unit
  // tiles g.  The FINALISER lane i of wave w' takes row 32 kq + 8 w' + (i >> 3), units 8 (i & 7) .. + 7 of the block's 64.
  float dc_reg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const uint32_t ring = smem_base + (uint32_t)wave * RING;
  const uint32_t dg_bytes = (uint32_t)((long)T * p.Btot * LDG * 2);
  const uint32_t rowoff = (uint32_t)p.row0;
  uint32_t dma_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * i + (lane >> 4);
    dma_off[i] = (uint32_t)((rowoff + r) * LDG * 2) + 2048u * kq + 512u * wave + (uint32_t)(((lane & 15) ^ r) << 4);
  }
  uint32_t frag_off[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) frag_off[kb] = (uint32_t)(n * 256 + (((4 * kb + q) ^ n) << 4));

  // [status 64 B | dflags [PNL][T][64]] as one buffer: flag word (l, t, j) at byte 64 + ((l T + t) 64 + j) 4
  const __amdgpu_buffer_rsrc_t r_fl = __builtin_amdgcn_make_buffer_rsrc(p.status, 0, 64 + PNL * T * 256, 0x00020000);
  const uint32_t lane4 = (uint32_t)(lane << 2);
  const uint32_t myfl = 64u + (uint32_t)(layer * T) * 256u, xfl = 64u + (uint32_t)((layer + 1) * T) * 256u;
  auto wait_my = [&](int tt) { return wait_flags_b(r_fl, myfl + (uint32_t)tt * 256u, lane4, p.spin_limit); };
  auto wait_x = [&](int tt) { return wait_flags_b(r_fl, xfl + (uint32_t)tt * 256u, lane4, p.spin_limit); };
  bool ok = true;

  // Slot list of a step; slot = (row tile ni = 0 .. 7, K-half kh), 4 KB.  ALL x slots (dG^{l+1}_t: the layer above is a diagonal ahead) come
  // first, then the h slots (dG^l_{t+1}: the layer's own previous step, published a moment ago):
  //     layers < top:  x(ni 0..7) | h(ni 0..3) -> half 0 complete | h(ni 4..7) -> half 1 complete          (32 slots)
  //     top layer:                  h(ni 0..3) -> half 0 complete | h(ni 4..7) -> half 1 complete          (16 slots)
  // so that 11 slots (2.3 us) of work that does not depend on the layer's own flags stand between a publish and the poll for the others'
  // (the first h slot is requested AHEAD = 5 slots before it is consumed) -- the flags' way through memory (1.5 us) is covered.  The price:
  // both halves' accumulators (128 registers) are live through the x phase.
  constexpr int NSLOT = HAS_X ? 32 : 16, NX = HAS_X ? 16 : 0, E0 = NX + 7, E1 = NX + 15;       // E0 / E1: last slot of half 0 / half 1
  constexpr int NST = 4;                           // stores of one half's K-quarter partial (per lane)
  constexpr int AHEAD = RS - 1;
  auto slot_is_x = [](int s) { return s < NX; };
  auto slot_ni = [](int s) { return (s % 16) / 2; };
  auto slot_kh = [](int s) { return s % 2; };
  const __amdgpu_buffer_rsrc_t rhs = __builtin_amdgcn_make_buffer_rsrc(p.dG[layer], 0, (int)dg_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(HAS_X ? p.dG[layer + 1] : p.dG[layer], 0, (int)dg_bytes, 0x00020000);
  uint32_t cpos = 0, ipos = 0;
  char* const myring = smem + wave * RING;
  auto issue1 = [&](int s, uint32_t sbase, int i) {
    const uint32_t so = sbase + (uint32_t)(slot_ni(s) * 16 * LDG * 2 + 256 * slot_kh(s));
    lds_void_t* d = (lds_void_t*)(myring + ipos + i * 1024);
    if (slot_is_x(s)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxs, d, 16, dma_off[i], so, 0, 16);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rhs, d, 16, dma_off[i], so, 0, 16);
  };
  auto advance = [](uint32_t& pos) { pos = (pos == (uint32_t)((RS - 1) * SLOT)) ? 0u : pos + (uint32_t)SLOT; };
  auto step_base = [&](int tt) -> uint32_t { return (uint32_t)((tt < 0 || tt >= T) ? 0 : tt) * (uint32_t)(p.Btot * LDG * 2); };
  // ---- K-slice exchange between the four waves + the lane re-mapping for the finaliser, in one trip through LDS.  A region holds one
  // 16-row x 64-unit fp32 tile as [row n][16 pieces of 4 units] x 16 B, piece column XOR-swizzled with the row (col ^ n): the writer (MFMA
  // layout: lane (n, q) holds piece 4 g + q of row n) and the reader (lane i' wants row 8 c + (i' >> 3), piece 2 (i' & 7) + pl -- what one
  // contiguous 1 KB store of the global exchange ring needs) are both bank-conflict free.  ALL four partials go through LDS (the own one
  // too), so every wave adds them in the order of the source wave, ((p0 + p1) + p2) + p3: row-position independent, no case distinctions.
  const uint32_t scr = smem_base + SCR_OFF;
  // (every lane offset below is derived from an opaque copy of the lane index where it is used: kept alive across the step, the dozen of
  // them would not fit beside 256 weight + 128 accumulator registers)
  auto opaque_lane = [&]() -> uint32_t { uint32_t l = (uint32_t)lane; asm volatile("" : "+v"(l)); return l; };
  auto write_partials = [&](f32x4 (&acc)[8][4], auto hc) {       // hc: which half's four row tiles
    constexpr int o = 4 * decltype(hc)::value;
    const uint32_t l = opaque_lane(), wn = l & 15, wq = l >> 4;
    const uint32_t xw_base = scr + (uint32_t)(wave << 12) + (wn * 256 + ((wq ^ (wn & 3)) << 4));       // + (d << 14) + ((g << 6) ^ xw_x)
    const uint32_t xw_x = (wn >> 2) << 6;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t a = xw_base + (((uint32_t)g << 6) ^ xw_x);
      asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc[o][g]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:16384" ::"v"(a), "v"(acc[o + 1][g]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:32768" ::"v"(a), "v"(acc[o + 2][g]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:49152" ::"v"(a), "v"(acc[o + 3][g]) : "memory");
    }
  };
  // Exchange ring: per (layer, parity of t, blk, dst, src) two planes of [32 rows][8 unit groups] float4: plane pl holds units 8 u8 + 4 pl .. + 3.
  const __amdgpu_buffer_rsrc_t r_ex = __builtin_amdgcn_make_buffer_rsrc(p.exch, 0, (int)EXCH_BYTES, 0x00020000);
  auto ex_off = [&](int par, int dst, int src) -> uint32_t { return (uint32_t)((((((layer * 2 + par) * 16 + blk) * 4 + dst) * 4 + src)) << 13); };
  // sum of the four K-slice partials of this wave's row tile, piece by piece in the finaliser's order, each piece straight out to the
  // workgroup that finalises these rows: half hf of wave w goes to dst = 2 hf + (w >> 1) (its rows 16 (w & 1) .. + 15)
  auto reduce_send_half = [&](int t, int hf) {
    const uint32_t ph = (uint32_t)(((T - 1 - t) >> 1) & 1);
    const uint32_t so = ex_off(t & 1, 2 * hf + (wave >> 1), kq);
    const uint32_t l = opaque_lane();
    const uint32_t xr_hi = scr + (uint32_t)(wave << 14) + (l >> 3) * 256;                      // + c * 2048 + (s << 12)
    const uint32_t xr_col = (2 * (l & 7)) ^ (l >> 3);                                          // ^ (pl + 8 c), << 4
    const uint32_t ex_wr = (uint32_t)((wave & 1) * 2048) + (l << 4);                           // + 4096 pl + 1024 c
    auto piece = [&](auto kc) {
      constexpr int k = decltype(kc)::value, c = k & 1, pl = k >> 1;
      const uint32_t a = xr_hi + ((xr_col ^ (uint32_t)(pl + 8 * c)) << 4);
      const f32x4 v0 = lds_rd<c * 2048>(a), v1 = lds_rd<c * 2048 + 4096>(a), v2 = lds_rd<c * 2048 + 8192>(a), v3 = lds_rd<c * 2048 + 12288>(a);
      wait_lgkmcnt<0>();
      u32x4_t v = __builtin_bit_cast(u32x4_t, ((v0 + v1) + v2) + v3);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] & ~1u) | ph;
      store_b128_wt(v, r_ex, ex_wr, so + (uint32_t)(4096 * pl + 1024 * c));
    };
    piece(std::integral_constant<int, 0>{}); piece(std::integral_constant<int, 1>{});
    piece(std::integral_constant<int, 2>{}); piece(std::integral_constant<int, 3>{});
    if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const __amdgpu_buffer_rsrc_t r_gates = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gates[layer]), 0, (int)((long)T * p.Btot * 4 * PH * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_cs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.cs[layer]), 0, (int)((long)T * p.Btot * PH * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_X ? reinterpret_cast<const float*>(p.cs[layer]) : p.dy), 0,
                                                                       (int)((long)T * p.Btot * PH * (HAS_X ? 2 : 4)), 0x00020000);

  if (HAS_X) ok = wait_x(T - 1);
  if (HAS_X) {
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue1(s, step_base(T - 1), i);
      advance(ipos);
    }
  }

  for (int t = T - 1; t >= 0 && ok; --t) {
    PBWD_STAMP(0);
    if (!HAS_X) {
      if (t < T - 1) ok = ok && wait_my(t + 1);
      PBWD_STAMP(2);
      ipos = cpos;
#pragma unroll
      for (int s = 0; s < AHEAD; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue1(s, step_base(t + 1), i);
        advance(ipos);
      }
    }
    const uint32_t xb = step_base(t), hb = step_base(t + 1), xb_next = step_base(t - 1);
    const uint32_t hmask = (t < T - 1) ? 0xffffffffu : 0u;
    f32x4 acc[8][4];
    // saved forward state of this lane's 8 cells: cold HBM reads, whole lines.  Requested in front of half 1's LDS exchange, where nothing
    // waits on the memory queue for the next ~2 us (the exchange, the partial stores, the partner workgroups' partials on their way).
    u32x4_t sg[4], sc_, scp, sdy[2];
    auto slot_body = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // flags the slot issued in THIS iteration (s + AHEAD) needs
      if constexpr (HAS_X && s + AHEAD == NX) {                   // the first h slot: the layer's own step t + 1
        if (t < T - 1) {
          PBWD_STAMP(1);
          ok = ok && wait_my(t + 1);
          PBWD_STAMP(2);
        }
      }
      if constexpr (HAS_X && s + AHEAD == NSLOT) {                // the next step's first x slot: the layer above's step t - 1
        if (t > 0) ok = ok && wait_x(t - 1);
      }
      constexpr int younger = (HAS_X ? (AHEAD - 1) : ((NSLOT - 1 - s) < (AHEAD - 1) ? (NSLOT - 1 - s) : (AHEAD - 1))) * 4;
      constexpr int stores_behind = (s > E0 && s <= E0 + AHEAD) ? NST : 0;
      wait_vmcnt<younger + stores_behind>();
      constexpr bool isx = s < NX;
      constexpr int ti = (s % 16) / 2, kh = s % 2;
      const uint32_t sb = ring + cpos;
      u32x4 b[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) b[kb] = lds_read128<0>(sb + frag_off[kb]);
      constexpr bool refill = HAS_X || (s + AHEAD < NSLOT);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        wait_lgkm(3 - kb);
        if (!isx) b[kb] &= hmask;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint4& w = isx ? wx[g][4 * kh + kb] : wh[g][4 * kh + kb];
          acc[ti][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, b[kb]), acc[ti][g], 0, 0, 0);
        }
        if constexpr (refill) {
          if constexpr (s + AHEAD < NSLOT) issue1(s + AHEAD, (s + AHEAD < NX) ? xb : hb, kb);
          else issue1(s + AHEAD - NSLOT, xb_next, kb);
        }
      }
      advance(cpos);
      if constexpr (refill) advance(ipos);
      if constexpr (s == E0) {
        PBWD_STAMP(3);
        write_partials(acc, std::integral_constant<int, 0>{});
        wait_lgkmcnt<0>();
        raw_barrier();
        reduce_send_half(t, 0);
      }
      if constexpr (s == E1) {
        PBWD_STAMP(4);
        raw_barrier();
        write_partials(acc, std::integral_constant<int, 1>{});
        {
          // (lane offsets derived here, from one opaque copy of the lane's row: kept alive across the step they would cost four registers)
          const uint32_t l = opaque_lane();
          const uint32_t row0 = rowoff + (uint32_t)(32 * kq + 8 * wave) + (l >> 3), u0 = (uint32_t)(64 * blk) + 8 * (l & 7);
          const uint32_t voff_g = (row0 * 4 * PH + u0) * 2, voff_c = (row0 * PH + u0) * 2, voff_dy = (row0 * PH + u0) * 4;
          const uint32_t hrow = (uint32_t)(t * p.Btot);
#pragma unroll
          for (int g = 0; g < 4; ++g) sg[g] = __builtin_amdgcn_raw_buffer_load_b128(r_gates, voff_g, hrow * (uint32_t)(4 * PH * 2) + (uint32_t)(g * PH * 2), 0);
          sc_ = __builtin_amdgcn_raw_buffer_load_b128(r_cs, voff_c, hrow * (uint32_t)(PH * 2), 0);
          const uint32_t prow = (uint32_t)((t > 0 ? t - 1 : 0) * p.Btot);
          scp = __builtin_amdgcn_raw_buffer_load_b128(r_cs, voff_c, prow * (uint32_t)(PH * 2), 0);
          if (!HAS_X) {
            sdy[0] = __builtin_amdgcn_raw_buffer_load_b128(r_dy, voff_dy, hrow * (uint32_t)(PH * 4), 0);
            sdy[1] = __builtin_amdgcn_raw_buffer_load_b128(r_dy, voff_dy, hrow * (uint32_t)(PH * 4) + 16u, 0);
          }
        }
        wait_lgkmcnt<0>();
        raw_barrier();
        reduce_send_half(t, 1);
      }
    };
    for_each_slot(slot_body, std::make_integer_sequence<int, NSLOT>{});
    PBWD_STAMP(5);
    // ---- the four K-quarter partials of this workgroup's 32 rows: polled until every word carries this use's phase bit
    u32x4_t part[4][2];
    {
      const uint32_t ex_rd = (opaque_lane() + (uint32_t)(wave << 6)) << 4;                     // + 4096 plane
      const uint32_t ph = (uint32_t)(((T - 1 - t) >> 1) & 1);
      bool got = false;
      for (uint32_t it = 0; it < p.spin_limit && !got; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) part[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(r_ex, ex_rd, ex_off(t & 1, kq, s) + (uint32_t)(4096 * pl), 16);
        uint32_t all1 = 1u, any1 = 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int e = 0; e < 4; ++e) { all1 &= part[s][pl][e]; any1 |= part[s][pl][e]; }
        const bool missing = ph ? ((all1 & 1u) == 0u) : ((any1 & 1u) != 0u);
        asm volatile("" ::: "memory");              // the loads are re-issued every round
        got = __builtin_amdgcn_ballot_w64(missing) == 0ull;
        if (!got) {
          if ((it & 63) == 63) {
            const uint32_t sv = __builtin_amdgcn_raw_buffer_load_b32(r_fl, 0u, 0u, 16);
            asm volatile("" ::: "memory");
            if (__builtin_amdgcn_readfirstlane(sv) != 0u) break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      ok = ok && got;
    }
    PBWD_STAMP(6);
    if (!ok) break;
    {
      const float cpm = t > 0 ? 1.f : 0.f;
      float dh[8], gi[8], gf[8], gg[8], go[8], c[8], cp[8], di[8], df[8], dg[8], dO[8];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const f32x4 v = ((__builtin_bit_cast(f32x4, part[0][pl]) + __builtin_bit_cast(f32x4, part[1][pl])) + __builtin_bit_cast(f32x4, part[2][pl])) +
                        __builtin_bit_cast(f32x4, part[3][pl]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dh[4 * pl + e] = v[e];
      }
      auto unpack8 = [](const u32x4_t v, float (&o)[8]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(v[k] << 16); o[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
      };
      unpack8(sg[0], gi); unpack8(sg[1], gf); unpack8(sg[2], gg); unpack8(sg[3], go); unpack8(sc_, c); unpack8(scp, cp);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = dh[e];
        if (!HAS_X) d += __uint_as_float(sdy[e >> 2][e & 3]);
        const float tc = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * c[e])), 1.f);
        const float d_o = d * tc;
        const float dc = d * go[e] * (1.f - tc * tc) + dc_reg[e];
        dc_reg[e] = dc * gf[e];
        di[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
        df[e] = dc * (cp[e] * cpm) * gf[e] * (1.f - gf[e]);
        dg[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
        dO[e] = d_o * go[e] * (1.f - go[e]);
      }
      auto pack8 = [](const float (&v)[8]) -> u32x4_t {
        u32x4_t r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (uint32_t)f2bf(v[2 * k]) | ((uint32_t)f2bf(v[2 * k + 1]) << 16);
        return r;
      };
      const uint32_t l = opaque_lane();
      const uint32_t row0 = rowoff + (uint32_t)(32 * kq + 8 * wave) + (l >> 3), u0 = (uint32_t)(64 * blk) + 8 * (l & 7);
      const uint32_t voff_dg = (row0 * LDG + u0) * 2;
      const uint32_t so = (uint32_t)(t * p.Btot) * (uint32_t)(LDG * 2);
      store_b128_wt(pack8(di), rhs, voff_dg, so);
      store_b128_wt(pack8(df), rhs, voff_dg, so + (uint32_t)(PH * 2));
      store_b128_wt(pack8(dg), rhs, voff_dg, so + (uint32_t)(2 * PH * 2));
      store_b128_wt(pack8(dO), rhs, voff_dg, so + (uint32_t)(3 * PH * 2));
    }
    PBWD_STAMP(7);
    // ---- publish dG^l_t
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(1u, r_fl, 0u, myfl + (uint32_t)(t * 256 + j * 4), 16);
  }
  if (!ok) {
    if (lane == 0) {
      if (atomicCAS(p.status, 0u, 1u) == 0u) { p.status[1] = blockIdx.x; p.status[2] = (uint32_t)layer; if (p.poison) *p.poison = __builtin_nanf(""); }
    }
  }
}

__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_kernel(PersistBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // a layer = one XCD pair (blocks b and b + 8 share an XCD); the four K-quarter partners of a unit block sit on ONE XCD
  const int xcd = blockIdx.x & 7, layer = xcd >> 1, r = blockIdx.x >> 3;        // r = 0 .. 31
  const int kq = r & 3, blk = ((r >> 2) << 1) | (xcd & 1);
  if (layer == PNL - 1) persist_bwd_body<false>(p, smem, layer, blk, kq);
  else persist_bwd_body<true>(p, smem, layer, blk, kq);
}

}  // namespace

// workspace: [status 64 B | dflags PNL x T x 64 words | exchange ring 16 MB, all ones at launch: phase bit 1, the first use of every slot writes phase 0]
size_t rnn_persist_bwd_workspace_bytes(int T) {
  size_t n = 64 + (size_t)PNL * T * 64 * 4;
  n = (n + 4095) & ~(size_t)4095;
  n += EXCH_BYTES;
#ifdef MVAE_TUNING
  n += (size_t)PNL * T * 64 * 8 * 8;       // clock samples of every workgroup (diagnostic build)
#endif
  return n;
}

bool rnn_persist_bwd_supported(const mvae_rnn_bwd_desc* d) {
  if (d->cell != MVAE_CELL_LSTM || d->dtype != MVAE_BF16 || d->layers != PNL || d->H != PH || (d->B != PB && d->B != 2 * PB) || d->T < 1) return false;
  if (!d->dy || d->dy_a || d->dy_ld != PH || d->lengths || d->drop_p > 0.f || d->ldg != LDG) return false;
  if (reinterpret_cast<uintptr_t>(d->dy) & 15) return false;
  for (int l = 0; l < PNL; ++l) {
    if (d->dh_last[l] || d->dh0[l] || !d->w_hhT[l] || !d->cs[l] || !d->gates[l] || !d->dG[l]) return false;
    if ((d->ldw_hhT[l] & 7) || (reinterpret_cast<uintptr_t>(d->w_hhT[l]) & 15) || (reinterpret_cast<uintptr_t>(d->dG[l]) & 15)) return false;
    if (l > 0 && (!d->w_ihT[l] || (d->ldw_ihT[l] & 7) || (reinterpret_cast<uintptr_t>(d->w_ihT[l]) & 15))) return false;
    if ((reinterpret_cast<uintptr_t>(d->gates[l]) & 15) || (reinterpret_cast<uintptr_t>(d->cs[l]) & 15)) return false;
  }
  if ((long)d->T * d->B * (long)LDG * 2 >= (1L << 31)) return false;      // 32-bit byte offsets / (int) buffer-resource extents in the kernel
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  return cus == PWG;
}

int rnn_persist_bwd(const mvae_rnn_bwd_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!rnn_persist_bwd_supported(d)) return MVAE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < rnn_persist_bwd_workspace_bytes(d->T) || (reinterpret_cast<uintptr_t>(ws) & 15)) return MVAE_ERR_WORKSPACE;
  PersistBwdArgs a;
  a.T = d->T; a.Btot = d->B; a.row0 = 0;
  for (int l = 0; l < PNL; ++l) {
    a.w_hhT[l] = d->w_hhT[l]; a.w_ihT[l] = d->w_ihT[l]; a.ldw_hhT[l] = d->ldw_hhT[l]; a.ldw_ihT[l] = d->ldw_ihT[l];
    a.gates[l] = d->gates[l]; a.cs[l] = d->cs[l]; a.dG[l] = d->dG[l];
  }
  a.dy = d->dy;
  const size_t flag_bytes = (size_t)PNL * d->T * 64 * 4;
  size_t head = 64 + flag_bytes;
  head = (head + 4095) & ~(size_t)4095;
  char* w = reinterpret_cast<char*>(ws);
  a.status = reinterpret_cast<uint32_t*>(w);
  a.poison = d->poison;
  mvae_tls_status = ws;
  a.dflags = reinterpret_cast<uint32_t*>(w + 64);
  a.exch = reinterpret_cast<float*>(w + head);
  a.dbg = nullptr;
#ifdef MVAE_TUNING
  a.dbg = reinterpret_cast<unsigned long long*>(w + head + EXCH_BYTES);
#endif
  const char* sp = mvae_knob("MVAE_PERSIST_SPIN");
  a.spin_limit = sp ? (uint32_t)atoi(sp) : (1u << 17);
  const char* sf = mvae_knob("MVAE_PERSIST_SAFE");
  a.safe = (sf && atoi(sf) != 0) ? 1 : 0;
  static std::atomic<bool> attr[MVAE_MAX_DEVICES];       // per-device function attribute
  int dev_id = 0;
  MVAE_CHECK_HIP(hipGetDevice(&dev_id));
  if (dev_id < 0 || dev_id >= MVAE_MAX_DEVICES || !attr[dev_id].load(std::memory_order_acquire)) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS));
    if (dev_id >= 0 && dev_id < MVAE_MAX_DEVICES) attr[dev_id].store(true, std::memory_order_release);
  }
  for (int r0 = 0; r0 < d->B; r0 += PB) {
    a.row0 = r0;
    if (r0 == 0) MVAE_CHECK_HIP(hipMemsetAsync(ws, 0, head, st));
    else MVAE_CHECK_HIP(hipMemsetAsync(w + 64, 0, flag_bytes, st));
    MVAE_CHECK_HIP(hipMemsetAsync(w + head, 0xff, EXCH_BYTES, st));       // no word carries phase 0 yet
    hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(PWG), dim3(256), PLDS, st, a);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
