// Epilogue of the 8-wave convolution kernels (conv_split3.hip: bf16x3 pieces; conv_h2.hip: fp16x2 pieces): the accumulators
// of a 256- / 128-row tile go through LDS to 16-byte row chunks -> scale (fp16x2 / per-level) + bias (+ residual) +
// activation -> global, or the raw partial tile of a split-K range; optionally the fused 1x1 head and the output's |max|.
#pragma once
#include "conv_split_common.hpp"

namespace odt {
namespace {

// |max| of a tensor as the kernels record it: the wave's maximum goes to the slot by one atomic (f32 bit patterns of
// non-negative values order like unsigned integers; NaNs drop out of fmaxf), skipped when the slot already holds more
__device__ __forceinline__ void publish_amax(unsigned* slot, float vmax, int tid) {
  if (slot == nullptr) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
  if ((tid & 63) == 0) {
    const unsigned b = __float_as_uint(vmax);
    unsigned* w = amax_way(slot);
    if (b > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, b);
  }
}

// ... once per workgroup: the waves' maxima meet in LDS (two barriers), thread 0 folds them into the slot -- an eighth of the
// slot reads, and of the same-address atomics of a layer's first round of tiles (every wave of it finds the slot empty)
template <int NTHR>
__device__ __forceinline__ void publish_amax_wg(unsigned* slot, float vmax, int tid, unsigned char* lds) {
  if (slot == nullptr) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
  float* red = reinterpret_cast<float*>(lds);
  ODT_BARRIER_LDS();                                   // the last pass of the C tile has been read
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  ODT_BARRIER_LDS();
  if (tid == 0) {
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NTHR / 64; ++w) m = fmaxf(m, red[w]);
    const unsigned b = __float_as_uint(m);
    unsigned* w = amax_way(slot);
    if (b > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, b);
  }
}

// Epilogue shared by the conv_split3 kernels: accumulators of the 8 waves (wave tile 64 x 32 TN at (wm, wn)) -> LDS ->
// rows of 16-byte chunks -> bias (+ residual) + activation -> global, or the raw partial tile of a split-K range.
template <int WM, int WN, int TN, int LDSB, bool TRACE, int NTHR = 512>
__device__ __forceinline__ void split3_epilogue(const ConvParams& p, f32x16 (&acc)[2][TN], unsigned char* lds, int m0, int n0,
                                                int M, int HoWo, int ks, int splitk, int tid, int wm, int wn, int fr, int fg,
                                                float h2_inv = 1.0f) {
  constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  // ---- epilogue: the C tile goes through LDS in passes of RP rows; per 16-byte row chunk: bias (+ residual)
  // + activation, 16-byte stores (a wave writes whole row segments).  The residual chunks of a pass are fetched
  // before the pass is staged, so their latency hides behind the LDS round trip.
  constexpr int CS = BN + 4;
  constexpr int FIT = LDSB / (CS * 4);                    // rows of the C tile the ring's LDS holds
  constexpr int RP = FIT >= BM ? BM : (FIT >= BM / 2 ? BM / 2 : (FIT >= BM / 4 ? BM / 4 : 64));   // rows per pass
  constexpr int NPASS = BM / RP, WPP = RP / 64;
  constexpr int C4 = BN / 4, RSTEP = NTHR / C4, NCH = RP / RSTEP;
  static_assert(RP >= 64 && BM % RP == 0 && RP % RSTEP == 0, "epilogue passes");
  float* Ct = reinterpret_cast<float*>(lds);
  const bool dense_io = p.out_oy == 0 && p.out_ox == 0 && p.out_H == p.Ho && p.out_W == p.Wo &&
                        (p.res_mode == 0 || (p.res_mode == 1 && p.res_H == p.Ho && p.res_W == p.Wo));
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.out, 0, (int)((unsigned)p.B * p.out_H * p.out_W * p.out_ldc * 4u), 0x00020000);
  // per-row-range constants (ConvParams::nlvl): every tile lies inside one range (ranges start on multiples of 256 rows)
  int lvl_off = 0;
  if (p.nlvl > 1) {
    // (constant indices: a runtime-indexed field would send the whole parameter record to scratch memory; unused entries
    // are INT_MAX)
    const int lvl = (m0 >= p.lvl_start[1] ? 1 : 0) + (m0 >= p.lvl_start[2] ? 1 : 0) + (m0 >= p.lvl_start[3] ? 1 : 0) +
                    (m0 >= p.lvl_start[4] ? 1 : 0);
    lvl_off = lvl * p.lvl_stride;
  }
  const unsigned nbias = p.nlvl > 1 ? (unsigned)(p.nlvl * p.lvl_stride) : (unsigned)p.Cout;
  const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)(nbias * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.lvl_scale != nullptr ? p.lvl_scale : p.bias), 0, (int)(p.lvl_scale != nullptr ? nbias * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res_mode != 0 ? p.res : p.bias), 0,
      (int)(p.res_mode != 0 ? (unsigned)p.B * p.res_H * p.res_W * p.res_ldc * 4u : 0u), 0x00020000);
  const int c4 = tid % C4, row0 = tid / C4;
  const int col = n0 + c4 * 4;
  if (splitk > 1) {
    // split-K: the raw partial tile, dense [M][Cout] rows of this range's slab (bias / residual / activation happen in
    // split_reduce_kernel once all ranges are in)
    const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.partial + (size_t)ks * M * cout_padded(p.Cout)), 0, (int)((unsigned)M * cout_padded(p.Cout) * 4u), 0x00020000);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      if (pass > 0) ODT_BARRIER_LDS();
      if (wm / WPP == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Ct[((wm % WPP) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg) * CS + wn * TN * 32 + j * 32 + fr] = acc[i][j][r];
      }
      ODT_BARRIER_LDS();
#pragma unroll
      for (int s2 = 0; s2 < NCH; ++s2) {
        const int m = m0 + pass * RP + row0 + s2 * RSTEP;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[(row0 + s2 * RSTEP) * CS + c4 * 4]);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_part, m < M ? (int)(((unsigned)m * cout_padded(p.Cout) + col) * 4u) : (int)kOOB, 0, 0);
      }
    }
    ODT_STAMP(5);
    return;
  }
  const f32x4 bias4 = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_bias, (lvl_off + col) * 4, 0, 0);
  // per-column factor in front of the bias: the per-level scale, or (fp16x2 pieces) the inverse of the weight column's and
  // the A operand's powers of two
  const bool has_scale = p.lvl_scale != nullptr || p.h2_chinv != nullptr;
  f32x4 scale4 = {1.f, 1.f, 1.f, 1.f};
  if (p.h2_chinv != nullptr) {
    const __amdgpu_buffer_rsrc_t rs_ch = __builtin_amdgcn_make_buffer_rsrc((void*)p.h2_chinv, 0, (int)((unsigned)cout_padded(p.Cout) * 4u), 0x00020000);
    scale4 = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_ch, col * 4, 0, 0) * h2_inv;
  } else if (has_scale) {
    scale4 = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_scale, (lvl_off + col) * 4, 0, 0);
  }
  if constexpr (BN == 256) {
    if (p.head_wt != nullptr) {
      // ---- fused 1x1 head (RPN class || box: 15 columns of a 16-wide GEMM over this tile's 256 channels).  Per pass:
      // accumulators -> LDS, bias + activation in place, then wave w multiplies rows [16 w, 16 w + 16) of the pass by
      // head_wt with v_mfma_f32_16x16x4_f32 (exact f32: an fmaf chain in k order).  k order of the chain: step
      // (t, u) takes channels 16 t + 4 j + u, j = lane / 16 -- one ds_read_b128 per lane feeds four MFMAs; the lane's 64
      // B-operand values (head_wt[16 t + 4 j + u][lane % 16]) are fetched once, up front.
      const int lane = tid & 63, wave = tid >> 6;
      const int hn = lane & 15, hj = lane >> 4;
      float hb[64];
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) hb[t * 4 + u] = p.head_wt[(16 * t + 4 * hj + u) * 16 + hn];
      const float hbias = p.head_bias[hn];
      static_assert(RP % 16 == 0, "head tiles");
      constexpr int RT = RP / 16;               // 16-row tiles per pass
#pragma unroll 1
      for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) ODT_BARRIER_LDS();
        if (wm / WPP == pass) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                Ct[((wm % WPP) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg) * CS + wn * TN * 32 + j * 32 + fr] = acc[i][j][r];
        }
        ODT_BARRIER_LDS();
#pragma unroll
        for (int s2 = 0; s2 < NCH; ++s2) {      // bias + activation in place (each thread its own 16-byte chunks)
          f32x4* q = reinterpret_cast<f32x4*>(&Ct[(row0 + s2 * RSTEP) * CS + c4 * 4]);
          f32x4 v = *q;
          if (has_scale) v = v * scale4;
          v += bias4;
          if (p.relu == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *q = v;
        }
        ODT_BARRIER_LDS();
        for (int rt = wave; rt < RT; rt += NTHR / 64) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
          const float* arow = &Ct[(rt * 16 + hn) * CS + 4 * hj];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], hb[t * 4 + u], c, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {          // C layout: row 4 j + i, column lane % 16
            const int m = m0 + pass * RP + rt * 16 + 4 * hj + i;
            if (m < M) p.head_out[(size_t)m * p.head_ldc + hn] = hn < 15 ? c[i] + hbias : 0.f;
          }
        }
      }
      ODT_STAMP(5);
      return;
    }
  }
  // (no barrier needed here: the last stage's barrier sits behind every fragment read of the ring)
  float vmax = 0.f;                          // |max| of what this thread stores (out_amax)
  const bool res_nt = (p.debug & 0x400) != 0;      // A/B: residual chunks (read once, by this workgroup only) with the non-temporal hint
  // A/B: outputs larger than the last-level cache (>= 128 MB: the 1024-channel res4 tensors, the res2 / P2 ones) stored non-temporally
  const bool out_nt = (p.debug & 0x1000) != 0 && (double)p.B * p.out_H * p.out_W * p.out_ldc * 4.0 >= 134217728.0;
  auto run = [&](auto act_c, auto res_c) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool RES = decltype(res_c)::value;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      unsigned ooff[NCH];
      f32x4 rres[RES ? NCH : 1];
#pragma unroll
      for (int s2 = 0; s2 < NCH; ++s2) {
        const int m = m0 + pass * RP + row0 + s2 * RSTEP;
        const bool ok = m < M;
        unsigned opix = (unsigned)m, rpix = (unsigned)m;
        if (!dense_io) {
          const int mm = ok ? m : 0;
          const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), rr = mm - n * HoWo;
          const int ho = sfast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
          opix = ((unsigned)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox;
          rpix = p.res_mode == 2 ? ((unsigned)n * p.res_H + (unsigned)(ho >> 1)) * p.res_W + (unsigned)(wo >> 1)
                                 : ((unsigned)n * p.res_H + (unsigned)ho) * p.res_W + (unsigned)wo;
        }
        ooff[s2] = ok ? (opix * p.out_ldc + col) * 4u : kOOB;
        if constexpr (RES)
          rres[s2] = res_nt ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? (int)((rpix * p.res_ldc + col) * 4u) : (int)kOOB, 0, 2)
                            : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? (int)((rpix * p.res_ldc + col) * 4u) : (int)kOOB, 0, 0);
      }
      if (pass > 0) ODT_BARRIER_LDS();        // the previous pass has been read
      if (wm / WPP == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Ct[((wm % WPP) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg) * CS + wn * TN * 32 + j * 32 + fr] = acc[i][j][r];
      }
      ODT_BARRIER_LDS();
      if (pass == 0) ODT_STAMP(3);
#pragma unroll
      for (int s2 = 0; s2 < NCH; ++s2) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[(row0 + s2 * RSTEP) * CS + c4 * 4]);
        if (has_scale) v = v * scale4;
        v += bias4;
        if constexpr (RES) v += rres[s2];
        if (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (ACT == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
        } else if (ACT == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
        }
        if (out_nt) __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)ooff[s2], 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)ooff[s2], 0, 0);
        if (ooff[s2] != kOOB) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
      if (pass == 0) ODT_STAMP(4);
    }
  };
  if (p.res_mode != 0) {
    if (p.relu == 1) run(std::integral_constant<int, 1>{}, std::true_type{});
    else if (p.relu == 0) run(std::integral_constant<int, 0>{}, std::true_type{});
    else if (p.relu == 2) run(std::integral_constant<int, 2>{}, std::true_type{});
    else run(std::integral_constant<int, 3>{}, std::true_type{});
  } else {
    if (p.relu == 1) run(std::integral_constant<int, 1>{}, std::false_type{});
    else if (p.relu == 0) run(std::integral_constant<int, 0>{}, std::false_type{});
    else if (p.relu == 2) run(std::integral_constant<int, 2>{}, std::false_type{});
    else run(std::integral_constant<int, 3>{}, std::false_type{});
  }
  if ((p.debug & 0x4000) != 0) publish_amax(p.out_amax, vmax, tid);       // A/B: one record per wave
  else publish_amax_wg<NTHR>(p.out_amax, vmax, tid, lds);
}

}  // namespace
}  // namespace odt
