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
//   * the four K-slice partials of the waves are exchanged through LDS (as in the forward pass), which leaves the workgroup with its
//     K-QUARTER partial of dh [128 x 64]; the four workgroups (blk, 0..3) exchange those through a 2-deep ring in global memory (write-through
//     16-byte stores into slots that hold a sentinel until then: the payload is its own flag): workgroup kq receives the 16 units 64 blk + 16 kq .. + 15 from all four, sums them in the
//     fixed order kq = 0, 1, 2, 3, and does the gate-derivative math for 128 rows x 16 units (dc carried in registers for the whole pass);
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
constexpr int SCR_OFF = 4 * RING, SCR_BYTES = 12 * 4096;
constexpr int PLDS = SCR_OFF + SCR_BYTES;         // 144 KB
constexpr size_t EXCH_BYTES = (size_t)PNL * 2 * 16 * 4 * 4 * 2 * 4096;      // [layer][parity of t][blk][dst kq][src kq][half][256 lanes] float4 = 16 MB

struct PersistBwdArgs {
  int T;
  int Btot, row0;
  const void* w_hhT[PNL]; const void* w_ihT[PNL]; long ldw_hhT[PNL]; long ldw_ihT[PNL];      // [H][ldw] bf16: row = hidden unit / input feature, column = gate row
  const float* dy;                 // [T][Btot][H] fp32: gradient w.r.t. the top layer's output
  const void* gates[PNL];          // [T][Btot][4H] bf16 (saved by the forward pass)
  const void* cs[PNL];             // [T][Btot][H] bf16
  void* dG[PNL];                   // [T][Btot][LDG] bf16 (out)
  float* exch;                     // EXCH_BYTES
  uint32_t* dflags;                // [PNL][T][2][64]: rows 64 hf .. + 63 of the dG^l_t tile of workgroup j are in memory
  uint32_t* status;
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

__device__ __forceinline__ void unpack4(const u32x2_t v, float (&o)[4]) {
  o[0] = __uint_as_float(v[0] << 16); o[1] = __uint_as_float(v[0] & 0xffff0000u);
  o[2] = __uint_as_float(v[1] << 16); o[3] = __uint_as_float(v[1] & 0xffff0000u);
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
  // The MFMA / LDS-exchange lane (n, q) holds batch rows 16 (4 hf + wave) + n, units 4 q .. + 3 of a 16-unit tile.  The FINALISER re-maps: lane
  // i takes row fr = i >> 2 and units 4 fq .. + 3, fq = i & 3 -- it simply reads exchange slot fr + 16 fq instead of its own, a transposition
  // that costs nothing -- so that four consecutive lanes cover the 32 contiguous bytes a row has per gate: 16 memory lines per load / store
  // instruction instead of 64 (saved gates, cell states, dG).
  const int fr = lane >> 2, fq = lane & 3;
  const int u0 = 64 * blk + 16 * kq + 4 * fq;
  float dc_reg[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

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

  uint32_t* myflags = p.dflags + (long)layer * T * 128;            // [t][half][64]
  const uint32_t* xflags = HAS_X ? p.dflags + (long)(layer + 1) * T * 128 : nullptr;
  bool ok = true;

  // slot list of a step as in the forward kernel: per 64-row half, x slots (dG^{l+1}_t) then h slots (dG^l_{t+1}); slot = (row tile ni, K-half kh)
  constexpr int NSLOT = HAS_X ? 32 : 16, HALF = NSLOT / 2;
  constexpr int NST = 4;                           // stores of one half's K-quarter partial (per lane)
  constexpr int AHEAD = RS - 1;
  auto slot_is_x = [](int s) { return HAS_X && (s % 16) < 8; };
  auto slot_ni = [](int s) { return HAS_X ? 4 * (s / 16) + (s % 8) / 2 : s / 2; };
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
  const uint32_t scr = smem_base + SCR_OFF;
  const uint32_t scr_lane = scr + (uint32_t)(lane << 4);
  auto write_partials = [&](f32x4 (&acc)[4][4], f32x4 (&own)[4]) {
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
    for (int g = 0; g < 4; ++g) own[g] = wave == 0 ? acc[0][g] : wave == 1 ? acc[1][g] : wave == 2 ? acc[2][g] : acc[3][g];
  };
  // sum of the four waves' K-slice partials in the order of the SOURCE wave (row-position independent, see rnn_persist.hip)
  auto reduce_half = [&](f32x4 (&tot)[4]) {
    const uint32_t rbase = scr_lane + (uint32_t)((wave * 3) << 12);
    f32x4 v[3][4];
    v[0][0] = lds_rd<0>(rbase); v[0][1] = lds_rd<1024>(rbase); v[0][2] = lds_rd<2048>(rbase); v[0][3] = lds_rd<3072>(rbase);
    v[1][0] = lds_rd<4096>(rbase); v[1][1] = lds_rd<5120>(rbase); v[1][2] = lds_rd<6144>(rbase); v[1][3] = lds_rd<7168>(rbase);
    v[2][0] = lds_rd<8192>(rbase); v[2][1] = lds_rd<9216>(rbase); v[2][2] = lds_rd<10240>(rbase); v[2][3] = lds_rd<11264>(rbase);
    wait_lgkmcnt<0>();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = tot[g];
      if (wave == 0) tot[g] = ((o + v[0][g]) + v[1][g]) + v[2][g];
      else if (wave == 1) tot[g] = ((v[2][g] + o) + v[0][g]) + v[1][g];
      else if (wave == 2) tot[g] = ((v[1][g] + v[2][g]) + o) + v[0][g];
      else tot[g] = ((v[0][g] + v[1][g]) + v[2][g]) + o;
    }
  };
  // K-quarter partial of one half -> the exchange ring: m-tile g goes to workgroup (blk, g) (the own tile too: every reader finds four
  // addends in one place and sums them in the same order, whoever it is)
  const __amdgpu_buffer_rsrc_t r_ex = __builtin_amdgcn_make_buffer_rsrc(p.exch, 0, (int)EXCH_BYTES, 0x00020000);
  const uint32_t ex_lane = (uint32_t)(tid << 4);
  auto ex_off = [&](int par, int dst, int src, int hf) -> uint32_t {
    return (uint32_t)(((((((layer * 2 + par) * 16 + blk) * 4 + dst) * 4 + src) * 2 + hf)) << 12);
  };
  auto send_half = [&](int t, int hf, const f32x4 (&tot)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, tot[g]), r_ex, ex_lane, ex_off(t & 1, g, kq, hf), 16);
    if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const __amdgpu_buffer_rsrc_t r_gates = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gates[layer]), 0, (int)((long)T * p.Btot * 4 * PH * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_cs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.cs[layer]), 0, (int)((long)T * p.Btot * PH * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_X ? reinterpret_cast<const float*>(p.cs[layer]) : p.dy), 0,
                                                                       (int)((long)T * p.Btot * PH * (HAS_X ? 2 : 4)), 0x00020000);
  const uint32_t row0 = rowoff + (uint32_t)(16 * wave + fr);
  const uint32_t ex_rd = (uint32_t)((64 * wave + fr + 16 * fq) << 4);      // the exchange slot this lane finalises
  const uint32_t voff_dg = (row0 * LDG + u0) * 2, voff_g = (row0 * 4 * PH + u0) * 2, voff_c = (row0 * PH + u0) * 2, voff_dy = (row0 * PH + u0) * 4;

  if (HAS_X) ok = wait_flags(xflags + (long)(T - 1) * 128, p.status, p.spin_limit, lane);
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
      if (t < T - 1) ok = ok && wait_flags(myflags + (long)(t + 1) * 128, p.status, p.spin_limit, lane);
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
    f32x4 acc[4][4], own0[4], own1[4];
    auto slot_body = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s % HALF == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // Flags are per 64-row HALF (the two halves of the batch are independent recurrences that take turns on this CU: while one half's
      // hand-off is on its way, the other half's slots are streamed).  The slot issued in THIS iteration (s + AHEAD) needs:
      if constexpr (HAS_X && s + AHEAD == 8) {                    // first h slot of half 0: the layer's own step t + 1, half 0
        if (t < T - 1) {
          PBWD_STAMP(1);
          ok = ok && wait_flags(myflags + (long)(t + 1) * 128, p.status, p.spin_limit, lane);
          PBWD_STAMP(2);
        }
      }
      if constexpr (HAS_X && s + AHEAD == 16) ok = ok && wait_flags(xflags + (long)t * 128 + 64, p.status, p.spin_limit, lane);      // first x slot of half 1
      if constexpr (s + AHEAD == (HAS_X ? 24 : 8)) {              // first h slot of half 1
        if (t < T - 1) ok = ok && wait_flags(myflags + (long)(t + 1) * 128 + 64, p.status, p.spin_limit, lane);
      }
      if constexpr (HAS_X && s + AHEAD == NSLOT) {                // the next step's first x slot (half 0)
        if (t > 0) ok = ok && wait_flags(xflags + (long)(t - 1) * 128, p.status, p.spin_limit, lane);
      }
      constexpr int younger = (HAS_X ? (AHEAD - 1) : ((NSLOT - 1 - s) < (AHEAD - 1) ? (NSLOT - 1 - s) : (AHEAD - 1))) * 4;
      constexpr int stores_behind = (s > HALF - 1 && s <= HALF - 1 + AHEAD) ? NST : 0;
      wait_vmcnt<younger + stores_behind>();
      constexpr bool isx = HAS_X && (s % 16) < 8;
      constexpr int ti = (HAS_X ? (s % 8) / 2 : (s / 2) % 4), kh = s % 2;
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
          if constexpr (s + AHEAD < NSLOT) issue1(s + AHEAD, ((s + AHEAD) % 16 < 8 && HAS_X) ? xb : hb, kb);
          else issue1(s + AHEAD - NSLOT, xb_next, kb);
        }
      }
      advance(cpos);
      if constexpr (refill) advance(ipos);
      if constexpr (s == HALF - 1) {
        PBWD_STAMP(3);
        write_partials(acc, own0);
        wait_lgkmcnt<0>();
        raw_barrier();
        reduce_half(own0);
        send_half(t, 0, own0);
      }
      if constexpr (s == NSLOT - 1) {
        PBWD_STAMP(4);
        raw_barrier();
        write_partials(acc, own1);
        wait_lgkmcnt<0>();
        raw_barrier();
        reduce_half(own1);
        send_half(t, 1, own1);
      }
    };
    for_each_slot(slot_body, std::make_integer_sequence<int, NSLOT>{});
    PBWD_STAMP(5);
    // ---- saved forward state of this lane's cells (both halves), requested behind the last partial stores
    u32x2_t sg[2][4], sc_[2], scp[2];
    u32x4_t sdy[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const uint32_t hrow = (uint32_t)(t * p.Btot + 64 * hf);
#pragma unroll
      for (int g = 0; g < 4; ++g) sg[hf][g] = __builtin_amdgcn_raw_buffer_load_b64(r_gates, voff_g, hrow * (uint32_t)(4 * PH * 2) + (uint32_t)(g * PH * 2), 0);
      sc_[hf] = __builtin_amdgcn_raw_buffer_load_b64(r_cs, voff_c, hrow * (uint32_t)(PH * 2), 0);
      const uint32_t prow = (uint32_t)((t > 0 ? t - 1 : 0) * p.Btot + 64 * hf);
      scp[hf] = __builtin_amdgcn_raw_buffer_load_b64(r_cs, voff_c, prow * (uint32_t)(PH * 2), 0);
      if (!HAS_X) sdy[hf] = __builtin_amdgcn_raw_buffer_load_b128(r_dy, voff_dy, hrow * (uint32_t)(PH * 4), 0);
    }
    // ---- the four K-quarter partials of this workgroup's 16 units: every exchange slot holds the SENTINEL (all ones: a NaN no arithmetic
    // produces) until its producer's 16-byte store lands, so the data is its own flag -- the producer neither drains nor raises anything, the
    // consumer polls the payload itself -- and the consumer puts the sentinel back once it has the values (the slot's next store comes two
    // steps later, behind two rounds of dG flags that this workgroup raises after draining the re-arming stores).
    const float cpm = t > 0 ? 1.f : 0.f;
    auto recv_half = [&](int hf, u32x4_t (&part)[4]) -> bool {
      bool got = false;
      for (uint32_t it = 0; it < p.spin_limit && !got; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) part[s] = __builtin_amdgcn_raw_buffer_load_b128(r_ex, ex_rd, ex_off(t & 1, kq, s, hf), 16);
        bool missing = false;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) missing = missing || (part[s][e] == 0xffffffffu);
        asm volatile("" ::: "memory");              // the loads are re-issued every round
        got = __builtin_amdgcn_ballot_w64(missing) == 0ull;
        if (!got) {
          if ((it & 63) == 63) {
            uint32_t sv;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(sv) : "v"(p.status) : "memory");
            if (__builtin_amdgcn_readfirstlane(sv) != 0u) break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      return got;
    };
    // re-arm the four slots, gate-derivative math of one half, dG stores (write-through): 8 stores per lane
    auto finish_half = [&](int hf, const u32x4_t (&part)[4]) {
      const u32x4_t ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
      for (int s = 0; s < 4; ++s) __builtin_amdgcn_raw_buffer_store_b128(ones, r_ex, ex_rd, ex_off(t & 1, kq, s, hf), 16);
      const f32x4 dhv = ((__builtin_bit_cast(f32x4, part[0]) + __builtin_bit_cast(f32x4, part[1])) + __builtin_bit_cast(f32x4, part[2])) + __builtin_bit_cast(f32x4, part[3]);
      float gi[4], gf[4], gg[4], go[4], c[4], cp[4], di[4], df[4], dg[4], dO[4];
      unpack4(sg[hf][0], gi); unpack4(sg[hf][1], gf); unpack4(sg[hf][2], gg); unpack4(sg[hf][3], go);
      unpack4(sc_[hf], c); unpack4(scp[hf], cp);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dh = dhv[e];
        if (!HAS_X) dh += __uint_as_float(sdy[hf][e]);
        const float tc = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * c[e])), 1.f);
        const float d_o = dh * tc;
        const float dc = dh * go[e] * (1.f - tc * tc) + dc_reg[hf][e];
        dc_reg[hf][e] = dc * gf[e];
        di[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
        df[e] = dc * (cp[e] * cpm) * gf[e] * (1.f - gf[e]);
        dg[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
        dO[e] = d_o * go[e] * (1.f - go[e]);
      }
      auto pack4 = [](const float (&v)[4]) -> u32x2_t {
        u32x2_t r = {(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
        return r;
      };
      const uint32_t so = (uint32_t)(t * p.Btot + 64 * hf) * (uint32_t)(LDG * 2);
      __builtin_amdgcn_raw_buffer_store_b64(pack4(di), rhs, voff_dg, so, 16);
      __builtin_amdgcn_raw_buffer_store_b64(pack4(df), rhs, voff_dg, so + (uint32_t)(PH * 2), 16);
      __builtin_amdgcn_raw_buffer_store_b64(pack4(dg), rhs, voff_dg, so + (uint32_t)(2 * PH * 2), 16);
      __builtin_amdgcn_raw_buffer_store_b64(pack4(dO), rhs, voff_dg, so + (uint32_t)(3 * PH * 2), 16);
      if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto raise = [&](int hf) {
      raw_barrier();
      if (tid == 0) {
        uint32_t one = 1u;
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(myflags + (long)t * 128 + hf * 64 + j), "v"(one) : "memory");
      }
    };
    u32x4_t part0[4], part1[4];
    ok = ok && recv_half(0, part0);
    PBWD_STAMP(6);
    if (!ok) break;
    finish_half(0, part0);
    // Half 1's partials are polled BEHIND half 0's stores in the memory queue (everything returns in issue order): once they are here, half
    // 0's dG tile is in memory -- its flag goes up without a drain of its own.
    ok = ok && recv_half(1, part1);
    if (!ok) break;
    raise(0);
    finish_half(1, part1);
    PBWD_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raise(1);
  }
  if (!ok) {
    if (lane == 0) {
      if (atomicCAS(p.status, 0u, 1u) == 0u) { p.status[1] = blockIdx.x; p.status[2] = (uint32_t)layer; }
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

// workspace: [status 64 B | dflags PNL x T x 2 x 64 words | exchange ring 16 MB, all ones at launch]
size_t rnn_persist_bwd_workspace_bytes(int T) {
  size_t n = 64 + (size_t)PNL * T * 128 * 4;
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
  }
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
  const size_t flag_bytes = (size_t)PNL * d->T * 128 * 4;
  size_t head = 64 + flag_bytes;
  head = (head + 4095) & ~(size_t)4095;
  char* w = reinterpret_cast<char*>(ws);
  a.status = reinterpret_cast<uint32_t*>(w);
  a.dflags = reinterpret_cast<uint32_t*>(w + 64);
  a.exch = reinterpret_cast<float*>(w + head);
  a.dbg = nullptr;
#ifdef MVAE_TUNING
  a.dbg = reinterpret_cast<unsigned long long*>(w + head + EXCH_BYTES);
#endif
  const char* sp = mvae_knob("MVAE_PERSIST_SPIN");
  a.spin_limit = sp ? (uint32_t)atoi(sp) : (1u << 20);
  const char* sf = mvae_knob("MVAE_PERSIST_SAFE");
  a.safe = (sf && atoi(sf) != 0) ? 1 : 0;
  static bool attr = false;
  if (!attr) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS));
    attr = true;
  }
  for (int r0 = 0; r0 < d->B; r0 += PB) {
    a.row0 = r0;
    if (r0 == 0) MVAE_CHECK_HIP(hipMemsetAsync(ws, 0, head, st));
    else MVAE_CHECK_HIP(hipMemsetAsync(w + 64, 0, flag_bytes, st));
    MVAE_CHECK_HIP(hipMemsetAsync(w + head, 0xff, EXCH_BYTES, st));       // every exchange slot armed
    hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(PWG), dim3(256), PLDS, st, a);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
