// ---------------------------------------------------------------------------------------------------------
// The ResNet stem in one kernel: conv0 (7x7 stride 2 + BN + ReLU, nn.py:860-896) and pool0 (3x3 stride 2 max over the
// top/left zero-padded map, nn.py:784-792), fp16x2 arithmetic (conv_split_common.hpp), bit-identical to the two launches
// it replaces (conv_h2_kernel<1, 2> on the plan's 7x1 form of conv0 + maxpool3x3s2_kernel): same products, same k order.
//   conv0 as the plan poses it: a 7 x 1 conv over rows of 8 pixels x 4 channels (K = 7 x 32; the 8th pixel and the 4th
//   channel carry zero weights), stride 2 pixels -- every input pixel is read by ~14 windows.  The generic kernel fetches,
//   splits and stores each of them (3.7 GB through the vector cache, 0.9 G conversions, 3.7 GB of ds_write per b=8 step)
//   and writes a 1.07 GB map that the pool kernel reads back: 0.76 + 0.31 ms.  Here:
//   * a workgroup owns 8 x 7 POOLED pixels = the 17 x 15 conv pixels under them (255 of the 256 MFMA rows; 1.14 x the conv
//     work for the windows shared with the neighbours) = a 39 x 36 patch of the padded frame: fetched ONCE, split ONCE into
//     two f16 planes in LDS (22 KB);
//   * the A fragment of (conv pixel, kh, pixel pair) is 16 contiguous bytes of a plane -- two pixels x four channels -- at
//     lane base + a compile-time constant per k16 step: no address arithmetic, no im2col, no ds_write in the loop;
//   * the whole weight image (conv_h2's, 64-wide n-tile: 7 stages x 8 KB) sits in LDS for the life of the workgroup, which is
//     PERSISTENT: one per CU, a contiguous range of tiles each (neighbouring tiles share their halo in L2);
//   * operands swapped (weights as A, pixels as B): lanes along the conv pixels, registers along the channels -- x 2^-s 2^-t_c
//     + bias, ReLU in registers, 16-byte stores to a [256][64 + 4] f32 tile in LDS (conv pixels outside the map: zeros, which
//     the post-ReLU maximum ignores like the reference's pad); each thread then folds 3 x 3 x 16 bytes and stores a pooled
//     pixel's 16 bytes -- one window element behind each of the NEXT tile's first MFMA steps.  The conv map never reaches HBM;
//   * the next tile's patch is fetched (12 registers) under the current tile's MFMAs and split into the patch buffer behind
//     them: two barriers per tile;
//   * MFMA rows take 4 x 4 blocks of conv pixels, not rows of them: the sixteen lanes of a ds_read_b128 group then touch sixteen
//     different 16-byte slots (row-major pixels: 2.6 LDS cycles per group, the tile's LDS time equalled its MFMA time).
// 149 KB of LDS, 207 VGPRs.  b = 8 @1080p: 0.37 ms against 0.76 + 0.31 (profiles/r04_stem_fusion_ab.txt).
// Reference ops: as conv_split.hip, elementwise.hip (maxpool3x3s2_kernel).
#include <atomic>
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

#define ODT_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)

struct StemCfg {
  static constexpr int PY = 8, PX = 7;                     // pooled pixels per tile
  static constexpr int CH = 2 * PY + 1, CW = 2 * PX + 1;   // conv pixels under them: 17 x 15 = 255
  static constexpr int PH = 2 * CH + 5, PW = 2 * CW + 6;   // patch of the padded frame: 39 x 36 pixels (4 channels)
  static constexpr int NPIX = PH * PW, PLANE = NPIX * 8;   // one f16 plane: 8 bytes per pixel
  static constexpr int NLD = (NPIX + 511) / 512;           // patch pixels per thread
  static constexpr int BKG = 64 * 16, BPL = 4 * BKG, STAGE_B = 2 * BPL, NSTG = 7;      // conv_h2's weight stages, 64-wide n-tile
  static constexpr int WOFF = 0, POFF = NSTG * STAGE_B, COFF = POFF + 2 * PLANE;
  static constexpr int CS = 68;                            // floats per row of the conv tile
  static constexpr int LDS = COFF + 256 * CS * 4;
  static_assert(CH * CW <= 256 && PW % 2 == 0 && COFF % 16 == 0 && LDS <= 160 * 1024, "stem tile");
};

__global__ void __launch_bounds__(512, 2) conv_stem_kernel(const ConvParams* __restrict__ pp) {
  using G = StemCfg;
  constexpr int PY = G::PY, PX = G::PX, CW = G::CW, PW = G::PW, NPIX = G::NPIX, PLANE = G::PLANE, NLD = G::NLD;
  constexpr int BKG = G::BKG, BPL = G::BPL, STAGE_B = G::STAGE_B, POFF = G::POFF, COFF = G::COFF, CS = G::CS;
  const ConvParams p = *pp;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fg = lane >> 5;
  const int Hp = p.in_Ha, Wp = p.in_Wa, Ho0 = p.Ho, Wo0 = p.Wo, Hq = p.out_H, Wq = p.out_W;
  const int tyn = (Hq + PY - 1) / PY, txn = (Wq + PX - 1) / PX, per_img = tyn * txn;
  const long ntiles = (long)p.B * per_img;
  // persistent workgroups: XCD x takes a contiguous share of the tile list, workgroup w of it a contiguous range
  int wg = (int)blockIdx.x;
  const int nwg = (int)gridDim.x;
  {
    const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int t_begin = (int)(ntiles * wg / nwg), t_end = (int)(ntiles * (wg + 1) / nwg);
  if (t_begin >= t_end) return;

  // ---- the weight image: 7 stages x 8 KB, global -> LDS once
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt_split, 0, G::NSTG * STAGE_B, 0x00020000);
#pragma unroll
  for (int i = 0; i < G::NSTG; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + G::WOFF + (i * 8 + wave) * 1024), 16,
                                             lane * 16 + (i * 8 + wave) * 1024, 0, 0, 0);

  const int sexp = h2_in_scale_exp(p);
  const float a_scale = pow2f(sexp), h2_inv = pow2f(-sexp);
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * Hp * Wp * 16u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.out, 0, (int)((unsigned)p.B * Hq * Wq * (unsigned)p.out_ldc * 4u), 0x00020000);

  // ---- the patch: thread -> pixels tid + 512 j of the 39 x 36 patch (row-major: 576 contiguous bytes per patch row)
  int l_pr[NLD], l_pc[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int q = tid + 512 * j;
    l_pr[j] = q / PW; l_pc[j] = q - l_pr[j] * PW;
  }
  f32x4 ga[NLD];
  auto tile_at = [&](int t, int& n, int& py0, int& px0) {
    n = t / per_img;
    const int r = t - n * per_img, ty = r / txn;
    py0 = ty * PY; px0 = (r - ty * txn) * PX;
  };
  auto load_patch = [&](int t) {
    int n, py0, px0;
    tile_at(t, n, py0, px0);
    // conv pixel (cy, cx) reads frame rows 2 cy .. 2 cy + 6, pixels 2 cx .. 2 cx + 7; the tile's first conv pixel is
    // (2 py0 - 1, 2 px0 - 1): the patch starts at frame (4 py0 - 2, 4 px0 - 2) (outside the frame: zeros, never used)
    const int iy0 = 4 * py0 - 2, ix0 = 4 * px0 - 2;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int y = iy0 + l_pr[j], x = ix0 + l_pc[j];
      const bool ok = tid + 512 * j < NPIX && (unsigned)y < (unsigned)Hp && (unsigned)x < (unsigned)Wp;
      const unsigned off = ok ? (((unsigned)n * Hp + y) * Wp + x) * 16u : kOOB;
      ga[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int q = tid + 512 * j;
      if (q < NPIX) {
        unsigned h0, l0, h1, l1;
        split2h(ga[j][0], ga[j][1], a_scale, h0, l0);
        split2h(ga[j][2], ga[j][3], a_scale, h1, l1);
        *reinterpret_cast<u32x2*>(lds + POFF + q * 8) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(lds + POFF + PLANE + q * 8) = u32x2{l0, l1};
      }
    }
  };

  // ---- fragments.  Pixels (B operand): lane (fr, fg) of row block t -> a conv pixel (cyl, cxl) of the tile, k16 step s =
  // (kh = s / 2, pixel pairs 2 (s % 2) + fg): 16 bytes at ((2 cyl + kh) PW + 2 cxl + 4 (s % 2) + 2 fg) 8
  // Which conv pixel an MFMA row holds is free: ds_read_b128 serves a wave in 16-lane groups ({0-3, 12-15, 20-27} and
  // {4-11, 16-19, 28-31} of each half), and with the patch rows 18 x 16 bytes apart a conv row's step is 4 sixteen-byte slots
  // (mod 16): row-major pixels put two conv rows of a group on the same slots (2.6 LDS cycles per group).  Instead a group
  // takes a 4 x 4 BLOCK of conv pixels -- slots 4 dy + dx: all sixteen -- for columns 0..11 of rows 0..15 (12 groups); the
  // other four take a 4 x 3 block of columns 12..14 plus four pixels of row 16 (1.19 cycles per group on average).
  int a_base[2], c_row[2];
  bool c_in[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bool gb = (fr >= 4 && fr < 12) || (fr >= 16 && fr < 20) || fr >= 28;      // second lane group of the half
    const int i = gb ? (fr < 12 ? fr - 4 : (fr < 20 ? fr - 8 : fr - 16)) : (fr < 4 ? fr : (fr < 16 ? fr - 8 : fr - 12));
    const int g = (wm * 2 + t) * 2 + (gb ? 1 : 0);
    int cyl, cxl;
    if (g < 12) { cyl = 4 * (g / 3) + (i >> 2); cxl = 4 * (g % 3) + (i & 3); }
    else if (i < 12) { cyl = 4 * (g - 12) + i / 3; cxl = 12 + i % 3; }
    else {
      // row 16: {3, 7, 11, -} | {0, 1, 2, 4} | {5, 6, 8, 9} | {10, 12, 13, 14} (the first set fills the slots its block leaves free)
      const int j = i - 12;
      cyl = 16;
      cxl = g == 12 ? 4 * j + 3 : (g == 13 ? j + (j == 3 ? 1 : 0) : (g == 14 ? 5 + j + (j >= 2 ? 1 : 0) : (j == 0 ? 10 : 11 + j)));
    }
    c_in[t] = cxl < CW;                       // (group 12's fourth extra: the idle MFMA row)
    if (!c_in[t]) cxl = 0;
    a_base[t] = POFF + ((2 * cyl) * PW + 2 * cxl + 2 * fg) * 8;
    c_row[t] = c_in[t] ? cyl * CW + cxl : 255;
  }
  // weights (A operand): lane (fr, fg) -> column wn 32 + fr, k-group 2 (s % 2) + fg of stage s / 2
  const int b_base = G::WOFF + fg * BKG + (wn * 32 + fr) * 16;
  // epilogue constants of this lane's 16 channels wn 32 + 8 g + 4 fg + e
  f32x4 sc[4], bs[4];
  {
    const __amdgpu_buffer_rsrc_t rs_ch = __builtin_amdgcn_make_buffer_rsrc((void*)p.h2_chinv, 0, (int)((unsigned)cout_padded(p.Cout) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)((unsigned)p.Cout * 4u), 0x00020000);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = wn * 32 + 8 * g + 4 * fg;
      sc[g] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_ch, c * 4, 0, 0) * h2_inv;
      bs[g] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_bias, c * 4, 0, 0);
    }
  }

  // ---- prologue: weights and the first patch in LDS, the second patch in flight
  load_patch(t_begin);
  store_patch();
  ODT_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();
  if (t_begin + 1 < t_end) load_patch(t_begin + 1);

  float vmax = 0.f;
  float* Ct = reinterpret_cast<float*>(lds + COFF);
  // ---- pool0: thread -> items tid, tid + 512 = (pooled pixel, 16-byte channel group) of the tile; the window's 3 x 3 conv
  // pixels sit at (2 pyl + dy, 2 pxl + dx) of the conv tile.  A tile's pooling runs UNDER THE NEXT TILE'S MFMAs (one window
  // element behind each of steps 1 .. 9, the stores behind step 10); the last tile's behind the loop.
  int pl_at[2], pl_py[2], pl_px[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + 512 * k, pix = item >> 4, c4 = item & 15, pyl = pix / PX, pxl = pix - pyl * PX;
    const bool any = item < PY * PX * 16;    // (no such item: reads the tile's first window, stores outside every map)
    pl_at[k] = any ? ((2 * pyl) * CW + 2 * pxl) * CS + c4 * 4 : 0;
    pl_py[k] = pyl; pl_px[k] = any ? pxl : (1 << 20);
  }
  f32x4 pmx[2];
  auto pool_read = [&](int e) {              // window element e = 3 dy + dx
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Ct + pl_at[k] + ((e / 3) * CW + e % 3) * CS);
      if (e == 0) pmx[k] = v;
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) pmx[k][c] = fmaxf(pmx[k][c], v[c]);
      }
    }
  };
  auto pool_store = [&](int n, int py0, int px0) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int py = py0 + pl_py[k], px = px0 + pl_px[k];
      const bool ok = py < Hq && px < Wq;
      if (ok) vmax = fmaxf(vmax, fmaxf(fmaxf(pmx[k][0], pmx[k][1]), fmaxf(pmx[k][2], pmx[k][3])));
      const unsigned off = ok ? ((((unsigned)n * Hq + py) * Wq + px) * (unsigned)p.out_ldc + ((tid + 512 * k) & 15) * 4u) * 4u : kOOB;
      __builtin_amdgcn_raw_buffer_store_b128((u32x4)pmx[k], rs_out, (int)off, 0, 0);
    }
  };
  int pn = 0, ppy0 = 0, ppx0 = 0;            // the previous tile (whose conv tile is in LDS)
  auto tile = [&](int t, auto HP) {
    constexpr bool has_prev = decltype(HP)::value;
    int n, py0, px0;
    tile_at(t, n, py0, px0);
    // ---- 14 k16 steps, every operand in LDS; the next step's fragments are read under this step's MFMAs
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 fa[2][2][2], fb[2][2];             // [buffer][piece][row block], [buffer][piece]
    auto rd = [&](int s, int buf) {
      const int so = ((s >> 1) * PW + 4 * (s & 1)) * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fb[buf][q] = *reinterpret_cast<const f16x8*>(lds + b_base + (s >> 1) * STAGE_B + q * BPL + (s & 1) * 2 * BKG);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[buf][q][i] = *reinterpret_cast<const f16x8*>(lds + a_base[i] + q * PLANE + so);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int s = 0; s < 2 * G::NSTG; ++s) {
      const int b = s & 1;
      if (s + 1 < 2 * G::NSTG) rd(s + 1, b ^ 1);
      ODT_FENCE();
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = ODT_MFMA_F16(fb[b][0], fa[b][1][i], acc[i]);      // lo * hi
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = ODT_MFMA_F16(fb[b][1], fa[b][0][i], acc[i]);      // hi * lo
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = ODT_MFMA_F16(fb[b][0], fa[b][0][i], acc[i]);      // hi * hi
      ODT_FENCE();
      if constexpr (has_prev) {
        if (s >= 1 && s <= 9) pool_read(s - 1);
        if (s == 10) pool_store(pn, ppy0, ppx0);
        ODT_FENCE();
      }
    }
    ODT_BARRIER_LDS();                       // every wave has read the patch and the previous tile's conv tile
    // ---- conv0's epilogue in registers -> the conv tile; pixels outside the map (and the idle row) are zeros
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cyl = c_row[i] / CW, cxl = c_row[i] - cyl * CW;
      const bool ok = c_in[i] && (unsigned)(cy0 + cyl) < (unsigned)Ho0 && (unsigned)(cx0 + cxl) < (unsigned)Wo0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        v = v * sc[g];
        v += bs[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? fmaxf(v[e], 0.f) : 0.f;
        *reinterpret_cast<f32x4*>(Ct + c_row[i] * CS + wn * 32 + 8 * g + 4 * fg) = v;
      }
    }
    if (t + 1 < t_end) store_patch();        // the next tile's patch (fetched under the MFMAs above)
    ODT_BARRIER_LDS();
    if (t + 2 < t_end) load_patch(t + 2);
    pn = n; ppy0 = py0; ppx0 = px0;
  };
  tile(t_begin, std::false_type{});
  for (int t = t_begin + 1; t < t_end; ++t) tile(t, std::true_type{});
#pragma unroll
  for (int e = 0; e < 9; ++e) pool_read(e);
  pool_store(pn, ppy0, ppx0);
  publish_amax_wg<512>(p.out_amax, vmax, tid, lds);
}

#undef ODT_FENCE

}  // namespace

// conv0 posed as the plan's 7 x 1 conv over 8-pixel x 4-channel rows, fp16x2 weight image with a 64-wide n-tile, ReLU,
// `out` = the POOLED map [B, out_H, out_W, out_ldc]
bool conv_stem_fits(const ConvParams& p) {
  return p.wt_split != nullptr && p.wt_split_kind == 2 && p.wt_split_bn == 64 && p.Cout == 64 && p.kh == 7 && p.kw == 1 && p.Cin == 32 &&
         p.in_ldc == 4 && p.stride == 2 && p.dil == 1 && p.pad_t == 0 && p.pad_l == 0 && p.relu == 1 && p.res_mode == 0 &&
         p.in2 == nullptr && p.splitk <= 1 && p.head_wt == nullptr && p.f_wt == nullptr && p.nlvl <= 1 && p.in_amax != nullptr &&
         p.h2_chinv != nullptr && p.H == p.in_Ha && p.W == p.in_Wa && p.in_Wa >= 2 * p.Wo + 6 && p.in_Ha >= 2 * p.Ho + 5 &&
         p.out_ldc >= 64 && p.out_ldc % 4 == 0;
}

int launch_conv_stem(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  ODT_CHECK(conv_stem_fits(p) && p.out != nullptr && p.out_H == (p.Ho + 1 - 3) / 2 + 1 && p.out_W == (p.Wo + 1 - 3) / 2 + 1 &&
            p.out_oy == 0 && p.out_ox == 0, "conv stem: unsupported shape");
  // one persistent workgroup per CU of the CURRENT device (handles on different devices / partitions differ; several
  // host threads may launch at once): a per-device table, each slot written once with a value that depends on the device
  // only (a benign double fill writes the same number)
  static std::atomic<int> cus[64];
  int dev_id = 0;
  ODT_HIP(hipGetDevice(&dev_id));
  int ncu = dev_id >= 0 && dev_id < 64 ? cus[dev_id].load(std::memory_order_relaxed) : 0;
  if (ncu == 0) {
    hipDeviceProp_t prop;
    ODT_HIP(hipGetDeviceProperties(&prop, dev_id));
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (dev_id >= 0 && dev_id < 64) cus[dev_id].store(ncu, std::memory_order_relaxed);
  }
  const long ntiles = (long)p.B * ((p.out_H + StemCfg::PY - 1) / StemCfg::PY) * ((p.out_W + StemCfg::PX - 1) / StemCfg::PX);
  const int cap = (p.debug >> 20) & 0x3ff;   // (test knob ODT_STEM_GRID through fuse_stem: fewer workgroups, several tiles each on small frames)
  const long want = cap > 0 && cap < ncu ? cap : ncu;
  const unsigned grid = (unsigned)(ntiles < want ? ntiles : want);
  hipLaunchKernelGGL(conv_stem_kernel, dim3(grid), dim3(512), 0, stream, dev);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
