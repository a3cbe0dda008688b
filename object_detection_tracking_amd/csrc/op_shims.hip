// Stand-alone op entry points of the C ABI (include/odt.h: odt_op_*, odt_nn_cosine): host pointers in and out, each
// runs exactly the kernels the forward uses -- what the staged parity tests call.
#include "odt_model.hpp"

using namespace odt;
#define g_err (::odt::last_error())

namespace {

template <typename T>
struct Tmp {   // RAII device temp for the stand-alone ops
  T* d = nullptr;
  size_t n = 0;
  int alloc(size_t count) { n = count; ODT_HIP(hipMalloc((void**)&d, (count ? count : 1) * sizeof(T))); return 0; }
  int put(const T* h) { ODT_HIP(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice)); return 0; }
  int get(T* h, size_t count) { ODT_HIP(hipMemcpy(h, d, count * sizeof(T), hipMemcpyDeviceToHost)); return 0; }
  int zero() { ODT_HIP(hipMemset(d, 0, (n ? n : 1) * sizeof(T))); return 0; }
  ~Tmp() { if (d) (void)hipFree(d); }
};

int set_dev(int device) {
  int n = 0;
  ODT_HIP(hipGetDeviceCount(&n));
  ODT_CHECK(device >= 0 && device < n, "no such device");
  ODT_HIP(hipSetDevice(device));
  knobs_reload();          // stand-alone ops (tests, tuning): the environment as it is at this call
  return 0;
}

}  // namespace

extern "C" {

int odt_nn_cosine(int device, const float* gallery, const int32_t* seg_offsets, int T, const float* dets,
                  int N, int D, double* cost) {
  ODT_CHECK(T >= 0 && N >= 0 && D > 0, "odt_nn_cosine: bad sizes");
  if (T == 0 || N == 0) return 0;
  ODT_CHECK(gallery && seg_offsets && dets && cost, "odt_nn_cosine: null argument");
  if (set_dev(device)) return 1;
  const int G = seg_offsets[T];
  ODT_CHECK(G > 0 && seg_offsets[0] == 0, "odt_nn_cosine: bad segment offsets");
  for (int t = 0; t < T; ++t) ODT_CHECK(seg_offsets[t + 1] > seg_offsets[t], "odt_nn_cosine: empty track gallery");
  // persistent per-device scratch + its own stream (no allocation, no null stream, no device-wide sync per call)
  static std::mutex mu;
  static std::map<int, std::unique_ptr<CosineCtx>> ctxs;
  std::lock_guard<std::mutex> lk(mu);
  const int dev = device;
  std::unique_ptr<CosineCtx>& cx = ctxs[dev];
  if (!cx) cx.reset(new CosineCtx());
  std::vector<const float*> gr(G), dr(N);
  for (int g = 0; g < G; ++g) gr[g] = gallery + (size_t)g * D;
  for (int j = 0; j < N; ++j) dr[j] = dets + (size_t)j * D;
  return cx->run(dev, gr.data(), G, seg_offsets, T, dr.data(), N, D, cost);
}

int odt_op_conv2d(int device, const float* in, int B, int H, int W, int Cin, const float* wt_hwio,
                  const float* bias, int kh, int kw, int Cout, int stride, int dil, int pad_t, int pad_l,
                  int Ho, int Wo, int oy, int ox, const float* res, int res_mode, int relu, float* out) {
  ODT_CHECK(in && wt_hwio && out, "odt_op_conv2d: null argument");
  if (set_dev(device)) return 1;
  const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * (Ho + oy) * (Wo + ox) * Cout;
  std::vector<float> w((size_t)Cout * kh * kw * Cin), bz(Cout, 0.f);
  for (int y = 0; y < kh; ++y) for (int x = 0; x < kw; ++x) for (int i = 0; i < Cin; ++i) for (int o = 0; o < Cout; ++o)
    w[(((size_t)o * kh + y) * kw + x) * Cin + i] = wt_hwio[(((size_t)y * kw + x) * Cin + i) * Cout + o];
  const int rH = res_mode == 2 ? (Ho + 1) / 2 : Ho, rW = res_mode == 2 ? (Wo + 1) / 2 : Wo;
  Tmp<float> di, dw, db, dr, dout;
  if (di.alloc(nin) || dw.alloc(w.size()) || db.alloc(Cout) || dout.alloc(nout) || dout.zero()) return 1;
  if (di.put(in) || dw.put(w.data()) || db.put(bias ? bias : bz.data())) return 1;
  if (res && res_mode) { if (dr.alloc((size_t)B * rH * rW * Cout) || dr.put(res)) return 1; }
  ConvParams p; std::memset(&p, 0, sizeof(p));
  p.in = di.d; p.wt = dw.d; p.bias = db.d; p.res = (res && res_mode) ? dr.d : nullptr; p.out = dout.d;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.in_ldc = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.kh = kh; p.kw = kw; p.stride = stride; p.dil = dil; p.pad_t = pad_t; p.pad_l = pad_l;
  p.out_H = Ho + oy; p.out_W = Wo + ox; p.out_oy = oy; p.out_ox = ox; p.out_ldc = Cout;
  p.res_mode = p.res ? res_mode : 0; p.res_H = rH; p.res_W = rW; p.res_ldc = Cout; p.relu = relu;
  p.in_Ha = H; p.in_Wa = W;
  Tmp<unsigned long long> tr;
  const bool trace = env_knob(K_CONV_TRACE).set;
  const int max_blocks = 1 << 16;
  if (trace) { if (tr.alloc((size_t)max_blocks * 16) || tr.zero()) return 1; p.trace = tr.d; }
  if (launch_conv(p, nullptr)) return 1;      // warm
  if (trace) { if (tr.zero()) return 1; }
  if (launch_conv(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (trace) {   // tuning aid: per-phase wall-clock (100 MHz) statistics over the workgroups
    std::vector<unsigned long long> t((size_t)max_blocks * 16);
    if (tr.get(t.data(), t.size())) return 1;
    unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
    double ph[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < max_blocks; ++b) {
      const unsigned long long* q = &t[(size_t)b * 16];
      if (q[0] == 0) continue;
      ++nb; if (q[0] < t0) t0 = q[0]; if (q[5] > t1) t1 = q[5];
      for (int i = 0; i < 5; ++i) ph[i] += (double)(q[i + 1] - q[i]);
    }
    printf("[conv trace] blocks=%d span=%.1f us | per block avg us: prologue %.2f  mainloop %.2f  res-issue+stage0 %.2f  "
           "pass0 lds->stores %.2f  pass1 %.2f | sum %.2f\n", nb, (t1 - t0) / 100.0, ph[0] / nb / 100, ph[1] / nb / 100,
           ph[2] / nb / 100, ph[3] / nb / 100, ph[4] / nb / 100, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / nb / 100);
    // concurrency: how many blocks are in the main loop at the midpoint of the launch
    const unsigned long long mid = t0 + (t1 - t0) / 2; int in_main = 0, in_epi = 0, in_pro = 0;
    for (int b = 0; b < max_blocks; ++b) {
      const unsigned long long* q = &t[(size_t)b * 16];
      if (q[0] == 0) continue;
      if (mid >= q[0] && mid < q[1]) ++in_pro; else if (mid >= q[1] && mid < q[2]) ++in_main; else if (mid >= q[2] && mid < q[5]) ++in_epi;
    }
    {   // first dispatch wave vs the rest
      double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
      double pa = 0, pb = 0, ma = 0, mb = 0; int na = 0, nb2 = 0;
      for (int b = 0; b < max_blocks; ++b) {
        const unsigned long long* q = &t[(size_t)b * 16];
        if (q[0] == 0) continue;
        double* sx = (q[0] - t0 < 500) ? sa : sb;
        sx[0] += (double)(q[6] - q[0]); sx[1] += (double)(q[7] - q[6]); sx[2] += (double)(q[1] - q[7]);
        if (q[0] - t0 < 500) { pa += (double)(q[1] - q[0]); ma += (double)(q[2] - q[1]); ++na; }
        else { pb += (double)(q[1] - q[0]); mb += (double)(q[2] - q[1]); ++nb2; }
      }
      printf("[conv trace] first wave (%d blocks): prologue %.2f us, mainloop %.2f us | later (%d blocks): prologue %.2f us, mainloop %.2f us\n",
             na, na ? pa / na / 100 : 0.0, na ? ma / na / 100 : 0.0, nb2, nb2 ? pb / nb2 / 100 : 0.0, nb2 ? mb / nb2 / 100 : 0.0);
      if (na && nb2)
        printf("[conv trace] prologue split (setup / first loads+lds / barrier): first wave %.2f / %.2f / %.2f us, later %.2f / %.2f / %.2f us\n",
               sa[0] / na / 100, sa[1] / na / 100, sa[2] / na / 100, sb[0] / nb2 / 100, sb[1] / nb2 / 100, sb[2] / nb2 / 100);
    }
    printf("[conv trace] at mid-launch: %d blocks in prologue, %d in main loop, %d in epilogue\n", in_pro, in_main, in_epi);
    {   // placement and per-CU concurrency: for every CU, the fraction of its busy time with 0 / 1 / 2 / 3+
        // resident workgroups inside the main loop (lockstep shows up as time with 0 in the loop)
      std::map<unsigned, std::vector<int>> cu_blocks;
      for (int b = 0; b < max_blocks; ++b) {
        const unsigned long long* q = &t[(size_t)b * 16];
        if (q[0] == 0) continue;
        const unsigned hw = (unsigned)q[8], xcc = (unsigned)q[9] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cu_blocks[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
      }
      double frac[4] = {0, 0, 0, 0}; double busy = 0;
      for (auto& kv : cu_blocks) {
        std::vector<std::pair<unsigned long long, int>> ev;   // (time, +1/-1) for main-loop occupancy
        unsigned long long lo = ~0ull, hi = 0;
        for (int b : kv.second) {
          const unsigned long long* q = &t[(size_t)b * 16];
          ev.push_back({q[1], +1}); ev.push_back({q[2], -1});
          if (q[0] < lo) lo = q[0]; if (q[5] > hi) hi = q[5];
        }
        std::sort(ev.begin(), ev.end());
        unsigned long long prev = lo; int n = 0;
        for (auto& e : ev) {
          frac[n > 3 ? 3 : n] += (double)(e.first - prev); prev = e.first; n += e.second;
        }
        frac[0] += (double)(hi - prev); busy += (double)(hi - lo);
      }
      printf("[conv trace] %zu CUs seen; time share per CU with k workgroups in the main loop: k=0 %.3f  k=1 %.3f  k=2 %.3f  k>=3 %.3f\n",
             cu_blocks.size(), frac[0] / busy, frac[1] / busy, frac[2] / busy, frac[3] / busy);
      // dispatch order on XCD 0: which CU did the first blocks land on
      printf("[conv trace] XCD0 first blocks -> (se,cu,tg): ");
      for (int b = 0; b < 8 * 40 && b < max_blocks; b += 8) {
        const unsigned long long* q = &t[(size_t)b * 16];
        if (q[0] == 0) break;
        const unsigned hw = (unsigned)q[8];
        printf("%u.%u.%u ", (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 16) & 0xf);
      }
      printf("\n");
    }
    fflush(stdout);
  }
  return dout.get(out, nout);
}

int odt_op_conv2d_cat(int device, const float* a, int B, int Ho, int Wo, int Ca, const float* b2, int Hb,
                      int Wb, int Cb, int stride_b, const float* wa, const float* wb, const float* bias,
                      int Cout, int relu, float* out) {
  ODT_CHECK(a && b2 && wa && wb && out, "odt_op_conv2d_cat: null argument");
  ODT_CHECK((Ho - 1) * stride_b < Hb && (Wo - 1) * stride_b < Wb, "odt_op_conv2d_cat: second input too small");
  if (set_dev(device)) return 1;
  std::vector<float> w((size_t)Cout * (Ca + Cb)), bz(Cout, 0.f);
  for (int o = 0; o < Cout; ++o) {
    for (int i = 0; i < Ca; ++i) w[(size_t)o * (Ca + Cb) + i] = wa[(size_t)i * Cout + o];
    for (int i = 0; i < Cb; ++i) w[(size_t)o * (Ca + Cb) + Ca + i] = wb[(size_t)i * Cout + o];
  }
  Tmp<float> da, db2, dw, dbias, dout;
  const size_t na = (size_t)B * Ho * Wo * Ca, nb = (size_t)B * Hb * Wb * Cb, nout = (size_t)B * Ho * Wo * Cout;
  if (da.alloc(na) || db2.alloc(nb) || dw.alloc(w.size()) || dbias.alloc(Cout) || dout.alloc(nout) || dout.zero()) return 1;
  if (da.put(a) || db2.put(b2) || dw.put(w.data()) || dbias.put(bias ? bias : bz.data())) return 1;
  ConvParams p; std::memset(&p, 0, sizeof(p));
  p.in = da.d; p.wt = dw.d; p.bias = dbias.d; p.out = dout.d;
  p.B = B; p.H = Ho; p.W = Wo; p.Cin = Ca; p.in_ldc = Ca; p.in_Ha = Ho; p.in_Wa = Wo;
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.kh = 1; p.kw = 1; p.stride = 1; p.dil = 1;
  p.out_H = Ho; p.out_W = Wo; p.out_ldc = Cout; p.relu = relu;
  p.in2 = db2.d; p.Cin2 = Cb; p.in2_ldc = Cb; p.in2_Ha = Hb; p.in2_Wa = Wb; p.in2_stride = stride_b;
  if (launch_conv(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  return dout.get(out, nout);
}

// conv2 (3x3, stride 1, 'SAME' for dilation dil, C -> C, bias, ReLU) -> conv3 (1x1, C -> C3, bias (+ residual), ReLU?) on the
// fp16x2 kernels, C = 64 / 128 / 256 (dil 1 or 2, C3 % 64 == 0): fuse = 1 as ONE conv_h2k_kernel launch with the fused tail (the plan's fuse_bottleneck_tails),
// fuse = 0 as the two launches the fused form replaces (conv_h2k_kernel -> [M,C] tensor + recorded range -> conv_h2_kernel).
int odt_op_bottleneck_tail(int device, const float* in, int B, int H, int W, int C, const float* w2_hwio, const float* b2,
                           int dil, const float* w3_io, const float* b3, int C3, const float* res, int relu3, int fuse,
                           float* out) {
  ODT_CHECK(in && w2_hwio && b2 && w3_io && b3 && out, "odt_op_bottleneck_tail: null argument");
  ODT_CHECK((C == 256 || C == 128 || C == 64) && C3 % 64 == 0 && C3 > 0 && (dil == 1 || dil == 2), "odt_op_bottleneck_tail: C = 64 / 128 / 256, C3 % 64 == 0, dil 1 or 2");
  if (set_dev(device)) return 1;
  const size_t M = (size_t)B * H * W;
  std::vector<float> w2((size_t)C * 9 * C), w3((size_t)C3 * C);
  for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x) for (int i = 0; i < C; ++i) for (int o = 0; o < C; ++o)
    w2[(((size_t)o * 3 + y) * 3 + x) * C + i] = w2_hwio[(((size_t)y * 3 + x) * C + i) * C + o];
  for (int i = 0; i < C; ++i) for (int o = 0; o < C3; ++o) w3[(size_t)o * C + i] = w3_io[(size_t)i * C3 + o];
  Tmp<float> di, dw2, db2, dw3, db3, dres, dmid, dout, img2, img3, imgf;
  Tmp<unsigned> amax;
  if (di.alloc(M * C) || dw2.alloc(w2.size()) || db2.alloc(C) || dw3.alloc(w3.size()) || db3.alloc(C3) || dmid.alloc(M * C) ||
      dout.alloc(M * C3) || dout.zero() || amax.alloc(4 * kAmaxWays) || amax.zero()) return 1;
  if (di.put(in) || dw2.put(w2.data()) || db2.put(b2) || dw3.put(w3.data()) || db3.put(b3)) return 1;
  if (res) { if (dres.alloc(M * C3) || dres.put(res)) return 1; }
  if (launch_tensor_amax(di.d, M * C, amax.d, nullptr)) return 1;
  ConvParams a; std::memset(&a, 0, sizeof(a));
  a.in = di.d; a.wt = dw2.d; a.bias = db2.d; a.out = dmid.d;
  a.B = B; a.H = H; a.W = W; a.Cin = C; a.in_ldc = C; a.in_Ha = H; a.in_Wa = W; a.Ho = H; a.Wo = W; a.Cout = C;
  a.kh = 3; a.kw = 3; a.stride = 1; a.dil = dil; a.pad_t = dil; a.pad_l = dil;
  a.out_H = H; a.out_W = W; a.out_ldc = C; a.relu = 1;
  a.wt_split_kind = 2; a.wt_split_bm = 256; a.wt_split_bn = C; a.wt_split_kwr = 1; a.splitk = 1;
  a.in_amax = amax.d; a.out_amax = amax.d + kAmaxWays; a.debug = 0x400;
  conv_prepare(a);
  if (img2.alloc((conv_split_weight_bytes(C, 9 * C) + 3) / 4) || conv_make_split_weights(a, img2.d, nullptr)) return 1;
  a.wt_split = img2.d; a.h2_chinv = conv_h2_chinv(img2.d, C, 9 * C);
  ConvParams b; std::memset(&b, 0, sizeof(b));
  b.in = dmid.d; b.wt = dw3.d; b.bias = db3.d; b.out = dout.d; b.res = res ? dres.d : nullptr;
  b.B = B; b.H = H; b.W = W; b.Cin = C; b.in_ldc = C; b.in_Ha = H; b.in_Wa = W; b.Ho = H; b.Wo = W; b.Cout = C3;
  b.kh = 1; b.kw = 1; b.stride = 1; b.dil = 1;
  b.out_H = H; b.out_W = W; b.out_ldc = C3; b.relu = relu3 ? 1 : 0;
  b.res_mode = res ? 1 : 0; b.res_H = H; b.res_W = W; b.res_ldc = C3;
  b.wt_split_kind = 2; b.wt_split_bm = 256; b.wt_split_bn = C3 % 256 == 0 ? 256 : (C3 % 128 == 0 ? 128 : 64); b.splitk = 1;
  if (b.wt_split_bn == 64) b.wt_split_bm = 128;
  b.in_amax = amax.d + kAmaxWays; b.out_amax = amax.d + 2 * kAmaxWays; b.debug = 0x400;
  conv_prepare(b);
  if (img3.alloc((conv_split_weight_bytes(C3, C) + 3) / 4) || conv_make_split_weights(b, img3.d, nullptr)) return 1;
  b.wt_split = img3.d; b.h2_chinv = conv_h2_chinv(img3.d, C3, C);
  Tmp<ConvParams> rec;
  if (rec.alloc(2)) return 1;
  if (fuse) {
    ODT_CHECK(conv_h2f_fusable(a, b), "odt_op_bottleneck_tail: this pair is not fusable");
    if (imgf.alloc((conv_h2f_weight_bytes(C3, C) + 3) / 4) || conv_make_h2f_weights(dw3.d, C3, C, imgf.d, nullptr)) return 1;
    a.f_wt = imgf.d; a.f_chinv = conv_h2f_chinv(imgf.d, C3, C); a.f_bias = db3.d; a.f_res = b.res; a.f_res_ldc = C3;
    a.f_out = dout.d; a.f_out_ldc = C3; a.f_cout = C3; a.f_relu = b.relu; a.f_out_amax = amax.d + 2 * kAmaxWays;
    a.out = nullptr; a.out_amax = nullptr;
    ConvParams recs[2] = {a, b};
    if (rec.put(recs)) return 1;
    if (launch_conv_split(a, rec.d, nullptr)) return 1;
  } else {
    ConvParams recs[2] = {a, b};
    if (rec.put(recs)) return 1;
    if (launch_conv_split(a, rec.d, nullptr) || launch_conv_split(b, rec.d + 1, nullptr)) return 1;
  }
  ODT_HIP(hipDeviceSynchronize());
  return dout.get(out, M * C3);
}

int odt_op_stem(int device, const float* frame_pad, int B, int Hp, int Wp, const float* w_hwio, const float* bias, int fuse,
                int grid, float* out) {
  ODT_CHECK(frame_pad && w_hwio && bias && out && B > 0 && Hp >= 11 && Wp >= 11, "odt_op_stem: null argument / frame too small");
  if (set_dev(device)) return 1;
  const int Ho0 = (Hp - 7) / 2 + 1, Wo0 = (Wp - 7) / 2 + 1, Wa = 2 * Wo0 + 8;      // (room for the 8th, zero-weight tap: plan_fpn.hip)
  const int Hq = (Ho0 + 1 - 3) / 2 + 1, Wq = (Wo0 + 1 - 3) / 2 + 1;
  // the plan's layouts: frames as [B, Hp, Wa, 4] (4th channel and the pad columns zero), conv0 as a 7 x 1 conv over 8-pixel x
  // 4-channel rows: virtual weights [64][7][32]
  std::vector<float> x((size_t)B * Hp * Wa * 4, 0.f), wv((size_t)64 * 7 * 32, 0.f);
  for (int b = 0; b < B; ++b) for (int y = 0; y < Hp; ++y) for (int xx = 0; xx < Wp; ++xx) for (int c = 0; c < 3; ++c)
    x[(((size_t)b * Hp + y) * Wa + xx) * 4 + c] = frame_pad[(((size_t)b * Hp + y) * Wp + xx) * 3 + c];
  for (int y = 0; y < 7; ++y) for (int xx = 0; xx < 7; ++xx) for (int c = 0; c < 3; ++c) for (int o = 0; o < 64; ++o)
    wv[((size_t)o * 7 + y) * 32 + xx * 4 + c] = w_hwio[(((size_t)y * 7 + xx) * 3 + c) * 64 + o];
  Tmp<float> di, dw, db, dmap, dout, img;
  Tmp<unsigned> amax;
  Tmp<ConvParams> rec;
  const size_t nmap = (size_t)B * Ho0 * Wo0 * 64, nout = (size_t)B * Hq * Wq * 64;
  if (di.alloc(x.size()) || dw.alloc(wv.size()) || db.alloc(64) || dmap.alloc(nmap) || dout.alloc(nout) || dout.zero() ||
      amax.alloc(4 * kAmaxWays) || amax.zero() || rec.alloc(1)) return 1;
  if (di.put(x.data()) || dw.put(wv.data()) || db.put(bias)) return 1;
  if (launch_tensor_amax(di.d, x.size(), amax.d, nullptr)) return 1;
  ConvParams p; std::memset(&p, 0, sizeof(p));
  p.in = di.d; p.wt = dw.d; p.bias = db.d; p.out = dmap.d;
  p.B = B; p.H = Hp; p.W = Wa; p.Cin = 32; p.in_ldc = 4; p.in_Ha = Hp; p.in_Wa = Wa; p.Ho = Ho0; p.Wo = Wo0; p.Cout = 64;
  p.kh = 7; p.kw = 1; p.stride = 2; p.dil = 1; p.out_H = Ho0; p.out_W = Wo0; p.out_ldc = 64; p.relu = 1;
  p.wt_split_kind = 2; p.wt_split_bm = 128; p.wt_split_bn = 64; p.splitk = 1;
  p.in_amax = amax.d; p.out_amax = amax.d + kAmaxWays;
  conv_prepare(p);
  if (img.alloc((conv_split_weight_bytes(64, 224) + 3) / 4) || conv_make_split_weights(p, img.d, nullptr)) return 1;
  p.wt_split = img.d; p.h2_chinv = conv_h2_chinv(img.d, 64, 224);
  if (fuse) {
    ODT_CHECK(conv_stem_fits(p), "odt_op_stem: shape not taken by the stem kernel");
    p.out = dout.d; p.out_H = Hq; p.out_W = Wq; p.stem_pool = 1;
    if (grid > 0) p.debug |= (grid & 0x3ff) << 20;
    if (rec.put(&p) || launch_conv_split(p, rec.d, nullptr)) return 1;
  } else {
    if (rec.put(&p) || launch_conv_split(p, rec.d, nullptr)) return 1;
    if (launch_maxpool3x3s2(dmap.d, B, Ho0, Wo0, 64, dout.d, Hq, Wq, nullptr)) return 1;
  }
  ODT_HIP(hipDeviceSynchronize());
  return dout.get(out, nout);
}

int odt_op_preprocess(int device, const void* frames, int dtype, int B, int H, int W, int pad_t, int pad_l,
                      int Hp, int Wp, float* out) {
  ODT_CHECK(frames && out, "odt_op_preprocess: null argument");
  if (set_dev(device)) return 1;
  const size_t nin = (size_t)B * H * W * 3 * (dtype == ODT_DTYPE_U8 ? 1 : 4);
  Tmp<unsigned char> di; Tmp<float> dout;
  if (di.alloc(nin) || di.put((const unsigned char*)frames) || dout.alloc((size_t)B * Hp * Wp * 4)) return 1;
  if (launch_preprocess(di.d, dtype, B, H, W, pad_t, pad_l, Hp, Wp, dout.d, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  return dout.get(out, dout.n);
}

int odt_op_maxpool(int device, const float* in, int B, int H, int W, int C, float* out) {
  ODT_CHECK(in && out, "odt_op_maxpool: null argument");
  if (set_dev(device)) return 1;
  const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
  Tmp<float> di, dout;
  if (di.alloc((size_t)B * H * W * C) || di.put(in) || dout.alloc((size_t)B * Ho * Wo * C)) return 1;
  if (launch_maxpool3x3s2(di.d, B, H, W, C, dout.d, Ho, Wo, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  return dout.get(out, dout.n);
}

int odt_op_topk(int device, const float* scores, int n, int k, int32_t* idx_out) {
  ODT_CHECK(scores && idx_out, "odt_op_topk: null argument");
  if (set_dev(device)) return 1;
  Tmp<float> ds; Tmp<int> di;
  if (ds.alloc(n) || ds.put(scores) || di.alloc(k)) return 1;
  if (launch_topk(ds.d, n, k, di.d, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  return di.get(idx_out, k);
}

int odt_op_nms(int device, const float* boxes, const float* scores, int n, int max_out, float iou_thresh,
               int32_t* idx_out, int* n_out) {
  ODT_CHECK(idx_out && n_out, "odt_op_nms: null argument");
  if (n == 0) { *n_out = 0; return 0; }
  ODT_CHECK(boxes && scores, "odt_op_nms: null argument");
  if (set_dev(device)) return 1;
  Tmp<float> db, ds; Tmp<int> di, dn;
  if (db.alloc((size_t)n * 4) || db.put(boxes) || ds.alloc(n) || ds.put(scores) || di.alloc(n) || dn.alloc(1)) return 1;
  if (launch_nms(db.d, ds.d, n, max_out, iou_thresh, di.d, dn.d, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (dn.get(n_out, 1)) return 1;
  return di.get(idx_out, *n_out);
}

int odt_op_proposals(int device, int graph, int B, int L, const int* hs, const int* ws, const int* fields,
                     const float* const* rpn, const float* const* anchors, int img_h, int img_w, int K,
                     float nms_thresh, float decode_clip, float* props, int32_t* nprops) {
  ODT_CHECK(L >= 1 && L <= 5 && hs && ws && fields && rpn && anchors && props && nprops, "odt_op_proposals: bad argument");
  if (set_dev(device)) return 1;
  ProposalParams p; std::memset(&p, 0, sizeof(p));
  Tmp<float> dr[5], da[5];
  for (int l = 0; l < L; ++l) {
    if (dr[l].alloc((size_t)B * hs[l] * ws[l] * kRpnCh) || dr[l].put(rpn[l])) return 1;
    if (da[l].alloc((size_t)fields[l] * fields[l] * 12) || da[l].put(anchors[l])) return 1;
    p.lvl[l].rpn = dr[l].d; p.lvl[l].anchors = da[l].d; p.lvl[l].h = hs[l]; p.lvl[l].w = ws[l]; p.lvl[l].field = fields[l];
  }
  p.nlevels = L; p.graph = graph; p.B = B; p.K = K; p.img_h = img_h; p.img_w = img_w;
  p.nms_thresh = nms_thresh; p.decode_clip = decode_clip;
  const size_t per = (size_t)B * L * K;
  Tmp<float> cb, cs, lb, ls, pr; Tmp<int> cc, lc, np;
  if (cb.alloc(per * 4) || cs.alloc(per) || lb.alloc(per * 4) || ls.alloc(per) || pr.alloc((size_t)B * K * 4) ||
      cc.alloc((size_t)B * L) || lc.alloc((size_t)B * L) || np.alloc(B)) return 1;
  p.cand_boxes = cb.d; p.cand_scores = cs.d; p.lvl_boxes = lb.d; p.lvl_scores = ls.d;
  p.cand_count = cc.d; p.lvl_count = lc.d; p.props = pr.d; p.nprops = np.d;
  Tmp<unsigned long long> ck;
  if (ck.alloc((size_t)B * proposal_total_chunks(p) * K)) return 1;
  p.chunk_keys = ck.d;
  if (launch_proposals(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (pr.get(props, (size_t)B * K * 4)) return 1;
  return np.get(nprops, B);
}

int odt_op_roi_align(int device, int B, int C, const int* hs, const int* ws, const float* const* feats,
                     const float* strides, const float* boxes, const int32_t* box_ind, int R,
                     float* out_nchw, float* pooled) {
  ODT_CHECK(hs && ws && feats && strides && boxes && box_ind && out_nchw, "odt_op_roi_align: null argument");
  if (R == 0) return 0;
  if (set_dev(device)) return 1;
  RoiAlignParams p; std::memset(&p, 0, sizeof(p));
  Tmp<float> df[4], db, dout, dpool; Tmp<int> di;
  for (int l = 0; l < 4; ++l) {
    if (df[l].alloc((size_t)B * hs[l] * ws[l] * C) || df[l].put(feats[l])) return 1;
    p.feat[l] = df[l].d; p.h[l] = p.alloc_h[l] = hs[l]; p.w[l] = p.alloc_w[l] = ws[l]; p.ldc[l] = C;
    p.inv_stride[l] = (float)(1.0 / (double)strides[l]);
  }
  if (db.alloc((size_t)R * 4) || db.put(boxes) || di.alloc(R) || di.put(box_ind) ||
      dout.alloc((size_t)R * C * 49) || dpool.alloc((size_t)R * C)) return 1;
  p.C = C; p.boxes = db.d; p.box_ind = di.d; p.per_image = 0; p.count = nullptr; p.R_cap = R;
  p.out_nchw = dout.d; p.pooled = pooled ? dpool.d : nullptr;
  if (launch_roi_align(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (dout.get(out_nchw, dout.n)) return 1;
  if (pooled) return dpool.get(pooled, dpool.n);
  return 0;
}

int odt_op_detections(int device, int graph, int B, int K, int C, const float* cls_logits,
                      const float* box_logits, const float* props, const int32_t* nprops, int img_h, int img_w,
                      const float* reg_weights, float decode_clip, float score_thresh, float nms_thresh,
                      int per_im, float* boxes, float* probs, int32_t* labels, int32_t* valid) {
  ODT_CHECK(cls_logits && box_logits && props && nprops && reg_weights && boxes && probs && labels && valid,
            "odt_op_detections: null argument");
  if (set_dev(device)) return 1;
  const int rows = B * K, ld = C * 5;
  std::vector<float> ho((size_t)rows * ld);
  for (int r = 0; r < rows; ++r) {
    std::memcpy(&ho[(size_t)r * ld], &cls_logits[(size_t)r * C], sizeof(float) * C);
    std::memcpy(&ho[(size_t)r * ld + C], &box_logits[(size_t)r * C * 4], sizeof(float) * C * 4);
  }
  DetectParams p; std::memset(&p, 0, sizeof(p));
  Tmp<float> dh, dp, dd, dpr, ob, op; Tmp<int> dn, ck, cc, ol, ov;
  if (dh.alloc(ho.size()) || dh.put(ho.data()) || dp.alloc((size_t)rows * 4) || dp.put(props) || dn.alloc(B) ||
      dn.put(nprops) || dd.alloc((size_t)rows * (C - 1) * 4) || dpr.alloc((size_t)rows * C) ||
      ck.alloc((size_t)B * (C - 1) * per_im) || cc.alloc((size_t)B * (C - 1)) || ob.alloc((size_t)B * per_im * 4) ||
      op.alloc((size_t)B * per_im) || ol.alloc((size_t)B * per_im) || ov.alloc(B)) return 1;
  p.graph = graph; p.B = B; p.K = K; p.C = C; p.head_out = dh.d; p.ld = ld; p.props = dp.d; p.nprops = dn.d;
  p.img_h = img_h; p.img_w = img_w;
  for (int i = 0; i < 4; ++i) p.reg_w[i] = reg_weights[i];
  p.decode_clip = decode_clip; p.score_thresh = score_thresh; p.nms_thresh = nms_thresh; p.per_im = per_im;
  p.dec_boxes = dd.d; p.probs = dpr.d; p.cls_keep = ck.d; p.cls_count = cc.d;
  p.out_boxes = ob.d; p.out_probs = op.d; p.out_labels = ol.d; p.out_valid = ov.d;
  if (launch_detections(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (ob.get(boxes, ob.n) || op.get(probs, op.n) || ol.get(labels, ol.n)) return 1;
  return ov.get(valid, B);
}

int odt_op_class_nms(int device, int graph, int B, int N, int C, const float* boxes_in, const float* scores_in,
                     const int32_t* ncand, float score_thresh, float nms_thresh, int per_im, float* boxes,
                     float* scores, int32_t* labels, int32_t* valid) {
  ODT_CHECK(boxes_in && scores_in && ncand && boxes && scores && labels && valid, "odt_op_class_nms: null argument");
  ODT_CHECK(C >= 1 && N >= 1, "odt_op_class_nms: bad sizes");
  if (set_dev(device)) return 1;
  const int rows = B * N, Cp = C + 1;
  std::vector<float> pr((size_t)rows * Cp, 0.f);
  for (int r = 0; r < rows; ++r) std::memcpy(&pr[(size_t)r * Cp + 1], &scores_in[(size_t)r * C], sizeof(float) * C);
  DetectParams p; std::memset(&p, 0, sizeof(p));
  Tmp<float> dd, dpr, ob, op; Tmp<int> dn, ck, cc, ol, ov;
  if (dd.alloc((size_t)rows * C * 4) || dd.put(boxes_in) || dpr.alloc(pr.size()) || dpr.put(pr.data()) || dn.alloc(B) ||
      dn.put(ncand) || ck.alloc((size_t)B * C * per_im) || cc.alloc((size_t)B * C) || ob.alloc((size_t)B * per_im * 4) ||
      op.alloc((size_t)B * per_im) || ol.alloc((size_t)B * per_im) || ov.alloc(B)) return 1;
  p.graph = graph; p.B = B; p.K = N; p.C = Cp; p.nprops = dn.d;
  p.score_thresh = score_thresh; p.nms_thresh = nms_thresh; p.per_im = per_im;
  p.dec_boxes = dd.d; p.probs = dpr.d; p.cls_keep = ck.d; p.cls_count = cc.d;
  p.out_boxes = ob.d; p.out_probs = op.d; p.out_labels = ol.d; p.out_valid = ov.d;
  if (launch_class_nms(p, nullptr)) return 1;
  ODT_HIP(hipDeviceSynchronize());
  if (ob.get(boxes, ob.n) || op.get(scores, op.n) || ol.get(labels, ol.n)) return 1;
  return ov.get(valid, B);
}

}  // extern "C"
