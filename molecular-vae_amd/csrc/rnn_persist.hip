// Weights-resident DATAFLOW recurrence for the per-rank shape of BASELINE configs[2] (b = 128 rows per GPU, 4 x LSTM(1024), bf16):
// nn.LSTM forward of models.py:156,164 as ONE persistent launch instead of T + 3 wavefront launches.
//
// Why: at b = 128 a wavefront launch is 7.3 GFLOP against 61 MB of weights that every launch must stream again (batch-independent), so the
// step kernels are bound by per-CU operand intake and by 123 dependent launches (16.8 us per diagonal, rnn.hip).  Here the WEIGHTS never move:
//   * 256 workgroups (one per CU, 4 waves, one wave per SIMD, the whole register file), 64 per layer.  Workgroup (l, j) owns hidden units
//     16 j .. 16 j + 15 of layer l: the 64 gate rows {i,f,g,o} x 16 units of [W_ih | W_hh] = 256 KB of bf16 held IN REGISTERS for the whole
//     pass (wave w keeps the K-slices k in [256 w, 256 w + 256) of both halves as MFMA A-fragments: 256 VGPRs / AGPRs per lane).
//   * per time step a workgroup takes in x_t = h^{l-1}_t and h^l_{t-1} (128 rows x 1024 k each, 512 KB) -- each wave ITS OWN k-slice, by LDS-DMA
//     into a private 3-slot ring (no inter-wave hand-off on the operand path, counted vmcnt only) -- and contracts them against its resident
//     weights: 512 MFMA 16x16x32 per wave and step (weights = A operand, batch rows = B operand, so that a lane ends up holding all four gates
//     of 4 consecutive hidden units of one batch row);
//   * the four K-slice partial sums are exchanged through LDS (wave w finalises batch rows 16 w .. 16 w + 15 of each 64-row half), the cell
//     update runs in registers (cell state c stays in registers for the whole pass), h / c / gates go out as 8-byte vectors;
//   * DATAFLOW between workgroups instead of launches or a grid barrier: h_t tiles are stored write-through (sc1), drained (vmcnt(0)), and the
//     workgroup raises its flag word flags[l][t][j]; a consumer (the same layer's 64 workgroups for step t + 1, the next layer's 64 for
//     step t) polls the 64 flag words of (l, t) with ONE 256-byte sc1 load per poll.  Every buffer is single-assignment inside the launch
//     (hs[l][t] is written once and read only after its flags), so no acquire fence is needed on the reading side.  The x-phase of step t
//     runs while the flags of the layer's own step t - 1 propagate.
// Spins are BOUNDED: a poll that does not succeed within `spin_limit` rounds writes an error record to `status` and every workgroup drains
// (the launch ends, results are garbage, the host sees status[0] != 0); nothing can hang.  All 256 workgroups must be co-resident: one
// per CU (512 registers per lane, 144 KB of LDS) on a 256-CU device -- the launcher refuses other devices.
#include "common.hpp"
#include "tile_pipe.hpp"
#include "kernels.hpp"
#include "persist_common.hpp"

namespace {

constexpr int PH = 1024, PNL = 4, PB = 128;        // hidden size, layers, batch rows of the one shape served
constexpr int LDH = PH + 64;                       // leading dimension of the hs buffers (elements): fixed, so that every offset is a literal
constexpr int PWG = 256;                          // workgroups: 64 per layer
constexpr int RS = 6;                             // ring slots per wave (4 KB each: 16 batch rows x 256 B = one row tile x HALF of the wave's K-slice)
constexpr int SLOT = 4096, RING = RS * SLOT;      // bytes
constexpr int SCR_OFF = 4 * RING;                 // reduction scratch behind the four rings: 12 regions x 4 KB
constexpr int SCR_BYTES = 12 * 4096;
constexpr int BIAS_OFF = SCR_OFF + SCR_BYTES;     // layers >= 1: the bias addend of every lane, [g][thread] float4 (16 KB) -- 16 registers freed
constexpr int PLDS = BIAS_OFF + 4 * 256 * 16;     // = 160 KB, all of the CU's LDS

struct PersistFwdArgs {
  int T;
  int Btot, row0;                  // the pass serves batch rows row0 .. row0 + 127 of buffers laid out for Btot rows per time step (B = 256: two passes)
  const void* w_ih[PNL]; const void* w_hh[PNL]; long ldw_ih[PNL]; long ldw_hh[PNL];
  const float* bias[PNL];          // [4H] fp32 (b_ih + b_hh) or null
  const float* add0;               // [B][4H] fp32: layer 0's time-invariant pre-activation (bias included) or null
  void* hs[PNL]; long ldh;         // [T][B][ldh] bf16
  void* cs[PNL];                   // [T][B][H] bf16 or null
  void* gates[PNL];                // [T][B][4H] bf16 or null
  float* cstate[PNL];              // [2][B][H] fp32: the final cell state goes to slot (T - 1) & 1
  uint32_t* flags;                 // [PNL][T][64], zero at launch
  uint32_t* status;                // [4]: error code, block, layer * 65536 + t, which poll
  float* poison;                   // optional: receives a quiet NaN when the launch gives up (mvae_rnn_fwd_desc.poison)
  uint32_t spin_limit;
  int safe;                        // != 0: every group of stores is drained (vmcnt(0)) right away -- the counted waits then never have a store in
                                   // flight (tests compare this form with the default one bit for bit)
  unsigned long long* dbg;         // diagnostic build (-DMVAE_TUNING) only: [PNL][T][8] clock samples of workgroup j = 0 of every layer, or null
};

#ifdef MVAE_TUNING
#define PERSIST_STAMP(K) do { if (p.dbg && j == 0 && tid == 0) p.dbg[((long)layer * T + t) * 8 + (K)] = wall_clock64(); } while (0)
#else
#define PERSIST_STAMP(K) do { } while (0)
#endif

// HAS_X: layers >= 1 (x part = the layer below); layer 0's input projection is time-invariant (add0).  SAVE: gates / cell states are written
// for the backward pass (6 stores per lane and half instead of 1).
template <bool HAS_X, bool SAVE>
__device__ __forceinline__ void persist_fwd_body(const PersistFwdArgs& p, char* smem, int layer, int j) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave index in an SGPR
  const int T = p.T, H = PH;
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;      // LDS byte address (address space 3 pointers are 32-bit offsets)
  const int n = lane & 15, q = lane >> 4;

  // ---- resident weights: A fragments, lane (m = n, kg = q) holds W[g * H + 16 j + m][256 w + 32 kb + 8 q .. + 8]
  uint4 wx[4][8], wh[4][8];
  {
    const bf16_t* wi = reinterpret_cast<const bf16_t*>(p.w_ih[layer]);
    const bf16_t* wr = reinterpret_cast<const bf16_t*>(p.w_hh[layer]);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const long row = (long)g * H + 16 * j + n;
        if (HAS_X) wx[g][kb] = *reinterpret_cast<const uint4*>(wi + row * p.ldw_ih[layer] + 256 * wave + 32 * kb + 8 * q);
        wh[g][kb] = *reinterpret_cast<const uint4*>(wr + row * p.ldw_hh[layer] + 256 * wave + 32 * kb + 8 * q);
      }
  }
  // lane (n, q) finalises batch rows  16 (4 hf + wave) + n  (hf = 0, 1), hidden units u0 .. u0 + 3
  const int u0 = 16 * j + 4 * q;
  float c_reg[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // the addend of the pre-activations, loaded ONCE (a load inside the step would make the compiler drain the operand ring in front of it).
  // Layers >= 1: b_ih + b_hh of this lane's units, parked in LDS (read back with ds_read_b128 in the cell update: the registers are needed
  // for the resident weights); layer 0 (no x weights: registers to spare): its time-invariant input projection, per finalised row.
  float addend0[HAS_X ? 1 : 2][4][4];
  const uint32_t bias_lds = smem_base + BIAS_OFF + (uint32_t)(tid << 4);
#pragma unroll
  for (int hf = 0; hf < (HAS_X ? 1 : 2); ++hf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_X) { if (p.bias[layer]) a = *reinterpret_cast<const float4*>(p.bias[layer] + g * H + u0); }
      else if (p.add0) a = *reinterpret_cast<const float4*>(p.add0 + (long)(p.row0 + 16 * (4 * hf + wave) + n) * 4 * H + g * H + u0);
      if (HAS_X) lds_write128(bias_lds + (uint32_t)(g * 4096), f32x4{a.x, a.y, a.z, a.w});
      addend0[hf][g][0] = a.x; addend0[hf][g][1] = a.y; addend0[hf][g][2] = a.z; addend0[hf][g][3] = a.w;
    }
  wait_lgkmcnt<0>();
  // ---- operand ring: slot = 16 batch rows x 256 bytes (4 k-blocks) of this wave's K-slice; LDS chunk position p of row r holds global chunk p ^ r
  const uint32_t ring = smem_base + (uint32_t)wave * RING;
  const uint32_t hs_bytes = (uint32_t)((long)T * p.Btot * LDH * 2);
  const uint32_t rowoff = (uint32_t)p.row0;          // first batch row of this pass
  // DMA: instruction i of a slot covers rows 4 i + (lane >> 4); the lane's LDS position (lane & 15) holds global chunk (lane & 15) ^ row:
  // every instruction reads whole 128-byte lines, the fragment reads below are bank-conflict free
  uint32_t dma_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * i + (lane >> 4);
    dma_off[i] = (uint32_t)((rowoff + r) * LDH * 2) + 512u * wave + (uint32_t)(((lane & 15) ^ r) << 4);
  }
  // fragment reads: lane (n, q), k-block kb of the slot: chunk (4 kb + q) ^ n of row n
  uint32_t frag_off[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) frag_off[kb] = (uint32_t)(n * 256 + (((4 * kb + q) ^ n) << 4));

  uint32_t* myflags = p.flags + (long)layer * T * 64;
  const uint32_t* xflags = HAS_X ? p.flags + (long)(layer - 1) * T * 64 : nullptr;
  bool ok = true;

  // Flat slot list of a step, one 64-row half after the other so that only ONE half's accumulators (64 registers) are live; a slot is
  // (row tile ni, K-half kh):
  //   layers >= 1:  x(ni 0..3) h(ni 0..3) | half 0 complete |  x(ni 4..7) h(ni 4..7) | half 1 complete          (32 slots)
  //   layer 0:                 h(ni 0..3) | half 0 complete |             h(ni 4..7) | half 1 complete          (16 slots)
  constexpr int NSLOT = HAS_X ? 32 : 16, HALF = NSLOT / 2;
  constexpr int NST = SAVE ? 6 : 1;                // stores of one half's cell update (per lane)
  constexpr int DEFER = 0;                         // half 0's exchange may be finished this many slots into half 1 (measured with 3: 1.395 vs
                                                   // 1.391 ms per pass -- the exchange is instruction issue, not latency: kept at 0)
  constexpr int AHEAD = RS - 1;                    // slot s + AHEAD is issued during iteration s, into the ring position iteration s - 1 read
  auto slot_is_x = [](int s) { return HAS_X && (s % 16) < 8; };
  auto slot_ni = [](int s) { return HAS_X ? 4 * (s / 16) + (s % 8) / 2 : s / 2; };
  auto slot_kh = [](int s) { return s % 2; };
  const __amdgpu_buffer_rsrc_t rhs = __builtin_amdgcn_make_buffer_rsrc(p.hs[layer], 0, (int)hs_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(HAS_X ? const_cast<void*>(p.hs[layer - 1]) : p.hs[layer], 0, (int)hs_bytes, 0x00020000);
  // Ring positions are RUNNING scalars (consume position cpos, issue position ipos = cpos + AHEAD mod RS), carried across steps: the slot
  // count of a step need not be a multiple of the ring depth.
  uint32_t cpos = 0, ipos = 0;                      // byte offsets inside this wave's ring
  char* const myring = smem + wave * RING;
  // ONE of the 4 LDS-DMA instructions of a slot -> ring position ipos.  sbase: byte offset of the source time step inside the hs buffer (a
  // scalar, computed once per step).  Slots that stand for zeros (h_{-1}) or for nothing (the fill for "step T" that keeps the number of
  // operations in flight the same in every step, so that every wait is ONE compile-time count) read valid memory that is never used.
  auto issue1 = [&](int s, uint32_t sbase, int i) {
    const uint32_t so = sbase + (uint32_t)(slot_ni(s) * 16 * LDH * 2 + 256 * slot_kh(s));
    lds_void_t* d = (lds_void_t*)(myring + ipos + i * 1024);
    if (slot_is_x(s)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxs, d, 16, dma_off[i], so, 0, 16);    // aux 16 = sc1
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rhs, d, 16, dma_off[i], so, 0, 16);
  };
  auto advance = [](uint32_t& pos) { pos = (pos == (uint32_t)((RS - 1) * SLOT)) ? 0u : pos + (uint32_t)SLOT; };
  auto step_base = [&](int tt) -> uint32_t { return (uint32_t)((tt < 0 || tt >= T) ? 0 : tt) * (uint32_t)(p.Btot * LDH * 2); };
  const uint32_t scr = smem_base + SCR_OFF;
  // Exchange of the K-slice partial sums: the tile wave d finalises is written by wave w != d to region 3 d + ((w - d) & 3) - 1 (4 KB each:
  // [gate][lane] float4), so that a reader finds its three addends in ONE contiguous 12 KB block (one base register, immediate offsets)
  // and sums them in a fixed order (source waves d + 1, d + 2, d + 3 mod 4: deterministic).
  const uint32_t scr_lane = scr + (uint32_t)(lane << 4);
  const __amdgpu_buffer_rsrc_t r_hs = rhs;
  const __amdgpu_buffer_rsrc_t r_gates = __builtin_amdgcn_make_buffer_rsrc(SAVE ? p.gates[layer] : p.hs[layer], 0, (int)((long)T * p.Btot * 4 * PH * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_cs = __builtin_amdgcn_make_buffer_rsrc(SAVE ? p.cs[layer] : p.hs[layer], 0, (int)((long)T * p.Btot * PH * 2), 0x00020000);
  const uint32_t row0 = rowoff + (uint32_t)(16 * wave + n);
  const uint32_t voff_h = (row0 * LDH + u0) * 2, voff_g = (row0 * 4 * PH + u0) * 2, voff_c = (row0 * PH + u0) * 2;
  auto write_partials = [&](f32x4 (&acc)[4][4], f32x4 (&own)[4]) {     // the three tiles other waves finalise -> scratch; own tile -> registers
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d == wave) continue;
      const uint32_t base = scr_lane + (uint32_t)((3 * d + ((wave - d) & 3) - 1) << 12);
      asm volatile("ds_write_b128 %0, %1" ::"v"(base), "v"(acc[d][0]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(base), "v"(acc[d][1]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(base), "v"(acc[d][2]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:3072" ::"v"(base), "v"(acc[d][3]) : "memory");
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) own[g] = wave == 0 ? acc[0][g] : wave == 1 ? acc[1][g] : wave == 2 ? acc[2][g] : acc[3][g];   // selects: no dynamic register index
  };
  typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_t;
  auto finish_half = [&](int t, int hf, f32x4 (&tot)[4], u32x2_t (&sv)[5]) {   // sum the other waves' partials (fixed order), cell update, h store
    const uint32_t rbase = scr_lane + (uint32_t)((wave * 3) << 12);
    {
      // v[k] = the partial of source wave (wave + 1 + k) & 3.  The four K-slice partials are added in the order of the SOURCE wave index,
      // ((p0 + p1) + p2) + p3, whichever wave finalises the tile: a molecule's result must not depend on the batch row it sits in
      // (test_bench_size_batch_tiled_from_the_reference_fixture: every copy of a molecule gives the same bits).
      f32x4 v[3][4];
      v[0][0] = lds_rd<0>(rbase); v[0][1] = lds_rd<1024>(rbase); v[0][2] = lds_rd<2048>(rbase); v[0][3] = lds_rd<3072>(rbase);
      v[1][0] = lds_rd<4096>(rbase); v[1][1] = lds_rd<5120>(rbase); v[1][2] = lds_rd<6144>(rbase); v[1][3] = lds_rd<7168>(rbase);
      v[2][0] = lds_rd<8192>(rbase); v[2][1] = lds_rd<9216>(rbase); v[2][2] = lds_rd<10240>(rbase); v[2][3] = lds_rd<11264>(rbase);
      wait_lgkmcnt<0>();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = tot[g];
        if (wave == 0) tot[g] = ((o + v[0][g]) + v[1][g]) + v[2][g];            // sources 0 (own) 1 2 3
        else if (wave == 1) tot[g] = ((v[2][g] + o) + v[0][g]) + v[1][g];       // sources 0 1 (own) 2 3
        else if (wave == 2) tot[g] = ((v[1][g] + v[2][g]) + o) + v[0][g];       // sources 0 1 2 (own) 3
        else tot[g] = ((v[0][g] + v[1][g]) + v[2][g]) + o;                      // sources 0 1 2 3 (own)
      }
    }
    f32x4 ad[4];
    if (HAS_X) {
      ad[0] = lds_rd<0>(bias_lds); ad[1] = lds_rd<4096>(bias_lds); ad[2] = lds_rd<8192>(bias_lds); ad[3] = lds_rd<12288>(bias_lds);
      wait_lgkmcnt<0>();
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) ad[g] = f32x4{addend0[HAS_X ? 0 : hf][g][0], addend0[HAS_X ? 0 : hf][g][1], addend0[HAS_X ? 0 : hf][g][2], addend0[HAS_X ? 0 : hf][g][3]};
    }
    // torch LSTM semantics, gates i f g o; row 16 (4 hf + wave) + n, units u0 .. u0 + 3
    float gi[4], gf[4], gg[4], go[4], hv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gi[e] = __builtin_amdgcn_rcpf(1.f + __expf(-(tot[0][e] + ad[0][e])));
      gf[e] = __builtin_amdgcn_rcpf(1.f + __expf(-(tot[1][e] + ad[1][e])));
      gg[e] = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * (tot[2][e] + ad[2][e]))), 1.f);
      go[e] = __builtin_amdgcn_rcpf(1.f + __expf(-(tot[3][e] + ad[3][e])));
      const float c = gf[e] * c_reg[hf][e] + gi[e] * gg[e];
      c_reg[hf][e] = c;
      hv[e] = go[e] * fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * c)), 1.f);
    }
    auto pack4 = [](const float (&v)[4]) -> u32x2_t {
      u32x2_t r = {(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
      return r;
    };
    // Buffer stores: ONE 32-bit lane offset per output stream (row 16 wave + n, units u0 ..), everything that varies with the step, the half
    // and the gate in the scalar offset -- 64-bit per-lane addresses for 6 streams x 2 halves would cost two dozen registers.
    const uint32_t hrow = (uint32_t)(t * p.Btot + 64 * hf);
    __builtin_amdgcn_raw_buffer_store_b64(pack4(hv), r_hs, voff_h, hrow * (uint32_t)(LDH * 2), 16);       // aux 16 = sc1: write-through, the hand-off payload
    if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SAVE) { sv[0] = pack4(gi); sv[1] = pack4(gf); sv[2] = pack4(gg); sv[3] = pack4(go); sv[4] = pack4(c_reg[hf]); }
  };
  // the saved state of one half (what the backward pass reads): 5 stores that nobody inside this launch waits for
  auto save_half = [&](int t, int hf, const u32x2_t (&sv)[5]) {
    const uint32_t hrow = (uint32_t)(t * p.Btot + 64 * hf);
    const uint32_t go_ = hrow * (uint32_t)(4 * PH * 2);
    __builtin_amdgcn_raw_buffer_store_b64(sv[0], r_gates, voff_g, go_, 0);
    __builtin_amdgcn_raw_buffer_store_b64(sv[1], r_gates, voff_g, go_ + (uint32_t)(PH * 2), 0);
    __builtin_amdgcn_raw_buffer_store_b64(sv[2], r_gates, voff_g, go_ + (uint32_t)(2 * PH * 2), 0);
    __builtin_amdgcn_raw_buffer_store_b64(sv[3], r_gates, voff_g, go_ + (uint32_t)(3 * PH * 2), 0);
    __builtin_amdgcn_raw_buffer_store_b64(sv[4], r_cs, voff_c, hrow * (uint32_t)(PH * 2), 0);
    if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // prologue: the first AHEAD slots of step 0 (x slots need the layer below's step 0; step 0's h slots are not used)
  if (HAS_X) ok = wait_flags(xflags, p.status, p.spin_limit, lane);
  if (HAS_X) {
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue1(s, 0u, i);
      advance(ipos);
    }
  }

  u32x2_t sv[5] = {};                               // half 1's saved state of the previous step (stored after the next step's h-flag poll)
  for (int t = 0; t < T && ok; ++t) {
    PERSIST_STAMP(0);
    if (!HAS_X) {
      // layer 0 has no x work to run ahead with: the step starts when the layer's own step t - 1 is complete everywhere
      if (t > 0) ok = ok && wait_flags(myflags + (long)(t - 1) * 64, p.status, p.spin_limit, lane);
      PERSIST_STAMP(3);
      if (SAVE && t > 0) save_half(t - 1, 1, sv);      // (see the h-flag poll of the other layers)
      ipos = cpos;
#pragma unroll
      for (int s = 0; s < AHEAD; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue1(s, step_base(t - 1), i);
        advance(ipos);
      }
    }
    // sources: x slots of step t read hs[l-1][t], h slots hs[l][t-1] (t = 0: any valid block, masked); the next step's first slots are x slots
    const uint32_t xb = step_base(t), hb = step_base(t - 1), xb_next = step_base(t + 1);
    const uint32_t hmask = (t > 0) ? 0xffffffffu : 0u;
    f32x4 acc[4][4], own0[4];
    // one iteration per slot, s a COMPILE-TIME constant (generic lambda over integral_constant: the unroller cannot decline)
    auto slot_body = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s % HALF == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // flags the slot issued in THIS iteration (s + AHEAD) needs: the first h slot of the step -> the layer's own step t - 1; the first slot
      // of the next step (an x slot) -> the layer below's step t + 1
      if constexpr (HAS_X && s + AHEAD == 8) {
        if (t > 0) {
          PERSIST_STAMP(2);
          ok = ok && wait_flags(myflags + (long)(t - 1) * 64, p.status, p.spin_limit, lane);
          PERSIST_STAMP(3);
          // the previous step's half-1 saved state goes out HERE: the poll has just drained the memory queue (nothing counted is in flight),
          // and the next poll is most of a step away -- issued right behind the publish, these stores' acknowledgements (~1 us) would be
          // what this poll waits for
          if (SAVE) save_half(t - 1, 1, sv);
        }
      }
      if constexpr (HAS_X && s + AHEAD == NSLOT) {
        if (t + 1 < T) {
          ok = ok && wait_flags(xflags + (long)(t + 1) * 64, p.status, p.spin_limit, lane);
          PERSIST_STAMP(1);
        }
      }
      // Slot s has landed when at most [DMA instructions of younger slots + stores issued behind slot s] operations are outstanding.
      // Layers >= 1 always have AHEAD - 1 younger slots in flight here (the next step's first slots, or the unused fill); layer 0 runs dry at
      // the end of a step.  The stores of half 0's cells go out at the end of iteration HALF - 1 + DEFER, behind that iteration's DMAs.
      constexpr int younger = (HAS_X ? (AHEAD - 1) : ((NSLOT - 1 - s) < (AHEAD - 1) ? (NSLOT - 1 - s) : (AHEAD - 1))) * 4;
      constexpr int stores_behind = (s > HALF - 1 + DEFER && s <= HALF - 1 + DEFER + AHEAD) ? NST : 0;
      wait_vmcnt<younger + stores_behind>();
      constexpr bool isx = HAS_X && (s % 16) < 8;
      constexpr int ti = (HAS_X ? (s % 8) / 2 : (s / 2) % 4), kh = s % 2;      // tile within the half, K-half
      const uint32_t sb = ring + cpos;
      u32x4 b[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) b[kb] = lds_read128<0>(sb + frag_off[kb]);
      constexpr bool refill = HAS_X || (s + AHEAD < NSLOT);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        wait_lgkm(3 - kb);
        if (!isx) b[kb] &= hmask;                  // step 0's h slots stand for h_{-1} = 0 (branch-free: a wave-uniform AND mask)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint4& w = isx ? wx[g][4 * kh + kb] : wh[g][4 * kh + kb];
          acc[ti][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, b[kb]), acc[ti][g], 0, 0, 0);
        }
        // one DMA instruction of slot s + AHEAD behind every k-block's MFMAs: the issue stall of the DMA runs under them.  Its ring position
        // is the one iteration s - 1 read (all of those fragment reads have returned).
        if constexpr (refill) {
          if constexpr (s + AHEAD < NSLOT) issue1(s + AHEAD, ((s + AHEAD) % 16 < 8 && HAS_X) ? xb : hb, kb);
          else issue1(s + AHEAD - NSLOT, xb_next, kb);
        }
      }
      advance(cpos);
      if constexpr (refill) advance(ipos);
      if constexpr (s == HALF - 1) {                // half 0 is complete: its partial sums go to the scratch now, the rest DEFER slots later
        PERSIST_STAMP(4);
        write_partials(acc, own0);
      }
      if constexpr (s == HALF - 1 + DEFER) {
        wait_lgkmcnt<0>();
        raw_barrier();
        u32x2_t sv0[5];
        finish_half(t, 0, own0, sv0);
        if (SAVE) save_half(t, 0, sv0);
        PERSIST_STAMP(5);
      }
      if constexpr (s == NSLOT - 1) {               // half 1 is complete: exchange, finish this wave's 16 rows, publish
        PERSIST_STAMP(6);
        raw_barrier();                              // everybody has finished READING half 0's partials
        f32x4 own[4];
        write_partials(acc, own);
        wait_lgkmcnt<0>();
        raw_barrier();
        finish_half(t, 1, own, sv);
      }
    };
    for_each_slot(slot_body, std::make_integer_sequence<int, NSLOT>{});
    PERSIST_STAMP(7);
    // ---- publish: every wave drains its h stores; then this workgroup's flag word for (layer, t) goes up.  Half 1's saved state follows
    // BEHIND the flag: nobody inside the launch waits for it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    if (tid == 0) {
      uint32_t one = 1u;
      asm volatile("global_store_dword %0, %1, off sc1" ::"v"(myflags + (long)t * 64 + j), "v"(one) : "memory");
    }
  }
  if (SAVE && ok) save_half(T - 1, 1, sv);
  if (!ok) {                                      // a bounded spin ran out (or somebody else's did): record who, drain
    if (lane == 0) {
      if (atomicCAS(p.status, 0u, 1u) == 0u) { p.status[1] = blockIdx.x; p.status[2] = (uint32_t)layer; if (p.poison) *p.poison = __builtin_nanf(""); }
    }
    return;
  }
  // final cell state (fp32), slot (T - 1) & 1 of the ping-pong the wavefront kernels keep
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int row = 16 * (4 * hf + wave) + n;
    *reinterpret_cast<float4*>(p.cstate[layer] + (long)((T - 1) & 1) * p.Btot * H + (long)(p.row0 + row) * H + u0) =
        make_float4(c_reg[hf][0], c_reg[hf][1], c_reg[hf][2], c_reg[hf][3]);
  }
}

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void lstm_persist_fwd_kernel(PersistFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // blocks b and b + 8 share an XCD (observed dispatch order; speed only): a layer = one XCD pair, so its 64 workgroups share two L2s
  const int xcd = blockIdx.x & 7, layer = xcd >> 1, j = ((blockIdx.x >> 3) << 1) | (xcd & 1);
  if (layer == 0) persist_fwd_body<false, SAVE>(p, smem, layer, j);       // ONE launch: all 256 workgroups must be co-resident
  else persist_fwd_body<true, SAVE>(p, smem, layer, j);
}

}  // namespace

// workspace: [status 64 B | flags PNL x T x 64 words]
size_t rnn_persist_fwd_workspace_bytes(int T) {
  size_t n = 64 + (size_t)PNL * T * 64 * 4;
#ifdef MVAE_TUNING
  n += (size_t)PNL * T * 8 * 8;                   // clock samples (diagnostic build)
#endif
  return n;
}

bool rnn_persist_fwd_supported(const mvae_rnn_fwd_desc* d) {
  // B = 128, or 256 as two passes over 128 rows each (the rows are independent: 2 x 1.4 ms against 3.4 ms of wavefront launches at B = 256)
  if (d->cell != MVAE_CELL_LSTM || d->dtype != MVAE_BF16 || d->layers != PNL || d->H != PH || (d->B != PB && d->B != 2 * PB) || d->T < 1) return false;
  if (d->x0 || d->lengths || d->add_table || d->add0_tstride != 0) return false;
  for (int l = 0; l < PNL; ++l) {
    if (d->h0[l] || d->hdrop[l] || !d->hs[l] || !d->w_hh[l] || !d->cstate[l]) return false;
    if (l > 0 && (!d->w_ih[l] || (d->ldw_ih[l] & 7))) return false;
    if ((d->ldw_hh[l] & 7) || (reinterpret_cast<uintptr_t>(d->w_hh[l]) & 15)) return false;
  }
  if (d->ldh != PH + 64) return false;
  // the kernel forms 32-bit byte offsets into hs / gates / cs and builds buffer resources of (int) extents: keep the largest per-layer
  // buffer ([T][B][4H] bf16) under 2 GiB, and every streamed / stored base 16-byte aligned
  if ((long)d->T * d->B * (4L * PH + 64) * 2 >= (1L << 31)) return false;
  for (int l = 0; l < PNL; ++l) {
    if (reinterpret_cast<uintptr_t>(d->hs[l]) & 15) return false;
    if (l > 0 && (reinterpret_cast<uintptr_t>(d->w_ih[l]) & 15)) return false;
    if ((d->cs[l] && (reinterpret_cast<uintptr_t>(d->cs[l]) & 15)) || (d->gates[l] && (reinterpret_cast<uintptr_t>(d->gates[l]) & 15))) return false;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  return cus == PWG;
}

// ws: rnn_persist_fwd_workspace_bytes(T) bytes of device memory, 16-byte aligned
int rnn_persist_fwd(const mvae_rnn_fwd_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!rnn_persist_fwd_supported(d)) return MVAE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < rnn_persist_fwd_workspace_bytes(d->T) || (reinterpret_cast<uintptr_t>(ws) & 15)) return MVAE_ERR_WORKSPACE;
  PersistFwdArgs a;
  a.T = d->T; a.Btot = d->B; a.row0 = 0;
  for (int l = 0; l < PNL; ++l) {
    a.w_ih[l] = d->w_ih[l]; a.w_hh[l] = d->w_hh[l]; a.ldw_ih[l] = d->ldw_ih[l]; a.ldw_hh[l] = d->ldw_hh[l];
    a.bias[l] = d->bias[l]; a.hs[l] = d->hs[l]; a.cs[l] = d->cs[l]; a.gates[l] = d->gates[l]; a.cstate[l] = d->cstate[l];
  }
  a.add0 = d->add0; a.ldh = d->ldh;
  a.status = reinterpret_cast<uint32_t*>(ws);
  a.poison = d->poison;
  mvae_tls_status = ws;
  a.flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + 64);
  a.dbg = nullptr;
#ifdef MVAE_TUNING
  a.dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + 64 + (size_t)PNL * d->T * 64 * 4);
#endif
  // ~0.2 s of polling rounds (2^17 x ~1.6 us): far beyond any step or any wait for a compute unit, short enough that a launch that cannot
  // make progress costs a fifth of a second, not two (round 5: a failure is survived, so the price of one is what matters).  MVAE_PERSIST_SPIN (tests: 1 = give up at the first
  // flag that is not there yet, which exercises the failure path) and MVAE_PERSIST_SAFE are honoured under MVAE_TUNING=1 only.
  const char* sp = mvae_knob("MVAE_PERSIST_SPIN");
  a.spin_limit = sp ? (uint32_t)atoi(sp) : (1u << 17);
  const char* sf = mvae_knob("MVAE_PERSIST_SAFE");
  a.safe = (sf && atoi(sf) != 0) ? 1 : 0;
  // the 160 KB dynamic-LDS opt-in is a per-DEVICE function attribute: once per device, not once per process
  static std::atomic<bool> attr[MVAE_MAX_DEVICES];
  int dev_id = 0;
  MVAE_CHECK_HIP(hipGetDevice(&dev_id));
  if (dev_id < 0 || dev_id >= MVAE_MAX_DEVICES || !attr[dev_id].load(std::memory_order_acquire)) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS));
    if (dev_id >= 0 && dev_id < MVAE_MAX_DEVICES) attr[dev_id].store(true, std::memory_order_release);
  }
  bool save = true;
  for (int l = 0; l < PNL; ++l) if (!d->gates[l] || !d->cs[l]) save = false;
  for (int r0 = 0; r0 < d->B; r0 += PB) {        // one pass per block of 128 rows (stream-ordered: the flags are re-zeroed in between)
    a.row0 = r0;
    // the status words survive between the passes of one call (a failure of the first pass must still be seen): only the flags are cleared
    if (r0 == 0) MVAE_CHECK_HIP(hipMemsetAsync(ws, 0, rnn_persist_fwd_workspace_bytes(d->T), st));
    else MVAE_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(ws) + 64, 0, (size_t)PNL * d->T * 64 * 4, st));
    if (save) hipLaunchKernelGGL(lstm_persist_fwd_kernel<true>, dim3(PWG), dim3(256), PLDS, st, a);
    else hipLaunchKernelGGL(lstm_persist_fwd_kernel<false>, dim3(PWG), dim3(256), PLDS, st, a);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
