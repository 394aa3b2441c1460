// Host side of libnlc_hip.so, model unit: upload of a NeuralLaplaceModel (weights re-packed into MFMA A-fragment order,
// nlc_pack.h), GRU encode, model forward (Fourier / de Hoog / linear algorithms) and the representation function alone.
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

namespace {
int64_t blob_size(const nlc_model_desc* d) {
  const int64_t g = d->h / 2, S = d->ilt.terms, P = d->d + 2, h = d->h;
  return 3 * g * d->nin + 3 * g * g + 6 * g + 3 * g * g + 3 * g * g + 6 * g + 2 * g + 2 + h * (2 * S + P) + h +
         h * h + h + 2 * d->d * S * h + 2 * d->d * S;
}
}  // namespace

// =================================================================================== model
extern "C" int64_t nlc_model_blob_size(const nlc_model_desc* d) { return d ? blob_size(d) : -1; }

extern "C" int nlc_set_model(nlc_ctx* c, const nlc_model_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (int r = check_ilt(c, &d->ilt)) return r;
  // (fixed_tablot / stehfest models: staged all-HIP forward and planner, like de Hoog)
  if (d->h != 64 && d->h != 128 && d->h != 256)
    return fail(c, NLC_ERR_UNSUPPORTED, "hidden_units must be 64, 128 or 256 (the kernels are instantiated for these widths)");
  if (d->d < 1 || d->d > 6) return fail(c, NLC_ERR_UNSUPPORTED, "state_dim must be in 1..6");
  if (d->nin < 1 || d->nin > NLC_MAX_NIN) return fail(c, NLC_ERR_UNSUPPORTED, "GRU input dim must be in 1..3");
  const bool linear_algo = d->ilt.algo == NLC_ILT_FIXED_TALBOT || d->ilt.algo == NLC_ILT_STEHFEST;
  if (!linear_algo && d->ilt.scale != 2.0) return fail(c, NLC_ERR_UNSUPPORTED, "fused model path needs ILT scale == 2");
  if (n != blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int g = d->h / 2, S = d->ilt.terms, P = d->d + 2, h = d->h, dd = d->d, nin = d->nin;
  Blob b{w, n};
  const double* Wih0 = b.take(3 * g * nin);
  const double* Whh0 = b.take(3 * g * g);
  const double* bih0 = b.take(3 * g);
  const double* bhh0 = b.take(3 * g);
  const double* Wih1 = b.take(3 * g * g);
  const double* Whh1 = b.take(3 * g * g);
  const double* bih1 = b.take(3 * g);
  const double* bhh1 = b.take(3 * g);
  const double* Wo = b.take(2 * g);
  const double* bo = b.take(2);
  const double* W1 = b.take((int64_t)h * (2 * S + P));
  const double* b1 = b.take(h);
  const double* W2 = b.take((int64_t)h * h);
  const double* b2 = b.take(h);
  const double* W3 = b.take((int64_t)2 * dd * S * h);
  const double* b3 = b.take(2 * dd * S);

  DeviceArena ar;
  // ---- GRU: layer-0 input weights with the bias folded into input column 3 (x = [a_0..a_{nin-1}, 0.., 1])
  std::vector<double> Wih0b((size_t)3 * g * 4, 0.0);
  for (int r = 0; r < 3 * g; ++r) {
    for (int j = 0; j < nin; ++j) Wih0b[(size_t)r * 4 + j] = Wih0[(size_t)r * nin + j];
    Wih0b[(size_t)r * 4 + 3] = bih0[r] + (r < 2 * g ? bhh0[r] : 0.0);
  }
  const size_t o_Wih0 = ar.push(pack_gru_chunked(Wih0b.data(), 4, 4, g));
  const size_t o_Whh0 = ar.push(pack_gru_chunked(Whh0, g, g, g));
  const size_t o_Wih1 = ar.push(pack_gru_chunked(Wih1, g, g, g));
  const size_t o_Whh1 = ar.push(pack_gru_chunked(Whh1, g, g, g));
  const size_t o_Wo = ar.push(pack_A(Wo, g, g, identity_rows(2)));
  std::vector<double> bhn0(bhh0 + 2 * g, bhh0 + 3 * g), brz1(2 * g), bin1(bih1 + 2 * g, bih1 + 3 * g),
      bhn1(bhh1 + 2 * g, bhh1 + 3 * g);
  for (int r = 0; r < 2 * g; ++r) brz1[r] = bih1[r] + bhh1[r];
  const size_t o_bhn0 = ar.push(bhn0), o_brz1 = ar.push(brz1), o_bin1 = ar.push(bin1), o_bhn1 = ar.push(bhn1);
  // the same hidden-state matrices as the int8 weight stream of kernels_gru_i8.hip (g == 64: one i8 MFMA covers K = 64)
  size_t o_i8 = 0;
  // (fixed point has no NaN / infinity, and rint of one is undefined: a model with a non-finite GRU weight or bias keeps the FP64 encoder)
  bool i8_ok = g == 64;
  for (int64_t i = 0; i8_ok && i < (int64_t)3 * g * g; ++i) i8_ok = std::isfinite(Whh0[i]) && std::isfinite(Wih1[i]) && std::isfinite(Whh1[i]);
  for (int i = 0; i8_ok && i < 3 * g; ++i) i8_ok = std::isfinite(bih1[i]) && std::isfinite(bhh1[i]) && std::isfinite(bhh0[i]);
  if (i8_ok) {
    const std::vector<signed char> st = pack_gru_i8_stream(Whh0, Wih1, Whh1, bhn0.data(), brz1.data(), bin1.data(), bhn1.data(), g);
    std::vector<double> v((st.size() + 7) / 8, 0.0);
    std::memcpy(v.data(), st.data(), st.size());
    o_i8 = ar.push(v);
  }

  // ---- representation MLP
  // layer 1 split: sphere-coordinate columns [0, 2S) and latent columns [2S, 2S+P)
  std::vector<double> W1s((size_t)h * 2 * S), W1p((size_t)h * 8, 0.0);
  for (int r = 0; r < h; ++r) {
    for (int j = 0; j < 2 * S; ++j) W1s[(size_t)r * 2 * S + j] = W1[(size_t)r * (2 * S + P) + j];
    for (int j = 0; j < P; ++j) W1p[(size_t)r * 8 + j] = W1[(size_t)r * (2 * S + P) + 2 * S + j];
  }
  const auto rowsh = identity_rows(h);
  const size_t o_W1s = ar.push(pack_A(W1s.data(), 2 * S, 2 * S, rowsh));
  const size_t o_W1p = ar.push(pack_A(W1p.data(), 8, 8, rowsh));
  const size_t o_b1 = ar.push(std::vector<double>(b1, b1 + h));
  const size_t o_W2 = ar.push(pack_A(W2, h, h, rowsh));
  const size_t o_b2 = ar.push(std::vector<double>(b2, b2 + h));
  // layer 3: slot layout + ILT coefficient matrix (nlc_pack.h)
  const int nt3 = nl_pick_nt3(ilt_tiles_needed(dd, S));
  if (nt3 < 0) return fail(c, NLC_ERR_UNSUPPORTED, "2*d*S too large for the fused kernel (max 25 output tiles)");
  const IltSlots slots = make_ilt_slots(dd, S, nt3);
  c->slot_elems = slots.elems;
  const int n_even_groups = slots.n_even_groups;
  std::vector<double> b3p((size_t)nt3 * 16, 0.0);
  for (size_t i = 0; i < slots.rowmap3.size(); ++i)
    if (slots.rowmap3[i] >= 0) b3p[i] = b3[slots.rowmap3[i]];
  const size_t o_W3 = ar.push(pack_A(W3, h, h, slots.rowmap3));
  const size_t o_b3 = ar.push(b3p);
  const size_t o_Cp = ar.push(slots.Cp);

  // ---- upload
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  {
    std::vector<int> slot((size_t)nt3 * 8, -1);
    for (size_t i = 0; i < slot.size(); ++i)
      if (slots.elems[i].first >= 0) slot[i] = slots.elems[i].first * S + slots.elems[i].second;
    if (c->slot_dev) hipFree(c->slot_dev);
    c->slot_dev = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->slot_dev, slot.size() * sizeof(int)));
    NLC_HIP(c, hipMemcpy(c->slot_dev, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> eidx((size_t)dd * S, 0);
    for (size_t i = 0; i < slot.size(); ++i)
      if (slot[i] >= 0) eidx[slot[i]] = (int)i;
    if (c->eidx_dev) hipFree(c->eidx_dev);
    c->eidx_dev = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->eidx_dev, eidx.size() * sizeof(int)));
    NLC_HIP(c, hipMemcpy(c->eidx_dev, eidx.data(), eidx.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->arena.base) hipFree(c->arena.base);
  c->arena = std::move(ar);
  c->arena.base = base;
  c->md = *d;
  c->g = g;
  c->S = S;
  c->P = P;
  c->W1s_host = std::move(W1s);
  c->b1_host.assign(b1, b1 + h);

  GruArgs& G = c->gru;
  G = GruArgs{};
  G.nin = nin;
  for (int j = 0; j < nin; ++j) {
    G.mean[j] = d->action_mean[j];
    G.std[j] = d->action_std[j];
  }
  G.Wih0p = base + o_Wih0;
  G.Whh0p = base + o_Whh0;
  G.Wih1p = base + o_Wih1;
  G.Whh1p = base + o_Whh1;
  G.Wop = base + o_Wo;
  G.bhn0 = base + o_bhn0;
  G.brz1 = base + o_brz1;
  G.bin1 = base + o_bin1;
  G.bhn1 = base + o_bhn1;
  G.bo[0] = bo[0];
  G.bo[1] = bo[1];
  if (i8_ok) G.i8_stream = (const signed char*)(base + o_i8);
  G.use_i8 = (i8_ok && c->opt_gru_gemm == 1) ? 1 : 0;  // (nlc_get_stat "gru_gemm" tells whether the option took)

  NlNetArgs& N = c->net;
  N = NlNetArgs{};
  N.d = dd;
  N.S = S;
  N.h = h;
  N.nt3 = nt3;
  N.n_even_groups = n_even_groups;
  N.W1p = base + o_W1p;
  N.W1s = base + o_W1s;
  N.b1 = base + o_b1;
  N.W2p = base + o_W2;
  N.b2 = base + o_b2;
  N.W3p = base + o_W3;
  N.b3p = base + o_b3;
  N.Cp = base + o_Cp;
  for (int i = 0; i < NLC_MAX_D; ++i) {
    N.state_mean[i] = i < dd ? d->state_mean[i] : 0.0;
    N.state_std[i] = i < dd ? d->state_std[i] : 1.0;
  }
  N.alpha = d->ilt.alpha;
  N.log_tol = std::log(d->ilt.tol);
  N.scale = d->ilt.scale;
  N.time_div = d->time_div;
  c->has_model = true;
  c->fwd_tn = -1.0;  // the constant-time forward's folded bias belongs to the previous weights
  c->has_mppi = false;  // a planner configured against the previous weights must be re-configured
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_gru_encode(nlc_ctx* c, const double* window, int64_t N, int B, double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!window || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  GruArgs a = c->gru;
  a.mode = 0;
  a.window = window;
  a.N = N;
  a.B = B;
  a.out = out;
  ProfScope ps(c, "gru_encode_kernel");
  NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int64_t nlc_model_workspace_bytes(nlc_ctx* c, int64_t N) {
  if (N < 0) return -1;
  int64_t n = N * 2 + 64;  // GRU latents
  if (c && c->has_model && c->md.ilt.algo == NLC_ILT_DEHOOG) n += 2 * N * c->md.d * c->S + 64;  // F_k re/im
  if (c && c->has_model && (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST))
    n += 2 * N * c->md.d * c->S + 2 * N * c->S + 128;  // theta / phi rows + per-row sphere inputs
  return n * (int64_t)sizeof(double);
}

extern "C" int nlc_model_forward_const_t(nlc_ctx* c, const double* obs, const double* window, double ts_pred, int64_t N,
                                         int B, double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (c->md.ilt.algo != NLC_ILT_FOURIER) return fail(c, NLC_ERR_UNSUPPORTED, "constant-time forward: fourier models only");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (!(ts_pred > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "ts_pred must be > 0");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  const double tn = ts_pred / c->md.time_div;
  const int h = c->md.h, S = c->S;
  if (!c->b1fold_fwd) NLC_HIP(c, hipMalloc((void**)&c->b1fold_fwd, h * sizeof(double)));
  if (tn != c->fwd_tn) {
    // fold the constant sphere inputs of layer 1 into its bias, as nlc_mppi_configure does for the planner
    // an earlier upload may still read the host copy, and a forward launched on a PREVIOUSLY bound stream may still read
    // the folded bias: a new query time is rare, so wait for the whole device
    NLC_HIP(c, hipDeviceSynchronize());
    std::vector<double> sph;
    sphere_inputs(c->md.ilt, tn, sph);
    c->fwd_fold_host.resize(h);
    for (int r = 0; r < h; ++r) {
      double acc = c->b1_host[r];
      for (int j = 0; j < 2 * S; ++j) acc += c->W1s_host[(size_t)r * 2 * S + j] * sph[j];
      c->fwd_fold_host[r] = acc;
    }
    NLC_HIP(c, hipMemcpyAsync(c->b1fold_fwd, c->fwd_fold_host.data(), h * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->fwd_tn = tn;
  }
  double* pa = (double*)ws;
  {
    GruArgs a = c->gru;
    a.mode = 0;
    a.window = window;
    a.N = N;
    a.B = B;
    a.out = pa;
    ProfScope ps(c, "gru_encode_kernel");
    NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  }
  ForwardArgs f{};
  f.net = c->net;
  f.net.b1 = c->b1fold_fwd;
  f.N = N;
  f.obs = obs;
  f.pa = pa;
  f.ts = nullptr;
  f.out = out;
  f.const_t = 1;
  f.tn = tn;
  ProfScope ps(c, "nl_forward_kernel");
  NLC_HIP(c, launch_nl_forward(f, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_model_forward(nlc_ctx* c, const double* obs, const double* window, const double* ts, int64_t N,
                                 int B, double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (c->md.ilt.algo == NLC_ILT_DEHOOG && (c->S < 3 || c->S > 33 || c->S % 2 == 0))
    return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !ts || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  double* pa = (double*)ws;
  {
    GruArgs a = c->gru;
    a.mode = 0;
    a.window = window;
    a.N = N;
    a.B = B;
    a.out = pa;
    ProfScope ps(c, "gru_encode_kernel");
    NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  }
  if (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST) {
    // staged (round 3): per-row query points s_k = node_k / t on the algorithm's own contour -> sphere inputs, the
    // representation kernel on explicit sphere inputs -> (theta, phi) rows, the Fourier kernel's stream with the algorithm's
    // per-term phase and weight
    const int S = c->S, dd = c->md.d;
    double* sph = pa + (N * 2 + 63) / 64 * 64;
    double* th = sph + (N * 2 * S + 63) / 64 * 64;
    double* ph = th + N * dd * S;
    const double* tab = nullptr;
    if (int r = linear_tables(c, &c->md.ilt, &tab)) return r;
    {
      RepInArgs ra{ts, ts, sph, N, 1, 0, S, 1, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale, tab, tab + S, c->md.time_div};
      ProfScope ps(c, "rep_inputs_kernel");
      NLC_HIP(c, launch_rep_inputs(ra, c->stream));
    }
    RepFuncArgs rf{};
    rf.net = c->net;
    rf.N = N;
    rf.obs = obs;
    rf.obs_stride = dd;
    rf.obs_per_sample = 1;
    rf.Kep = 1;
    rf.pa = pa;
    rf.pa_stride = 2;
    rf.general_t = 1;
    rf.slot = c->slot_dev;
    rf.fre = th;
    rf.fim = ph;
    rf.sph = sph;
    rf.sph_stride = 2 * S;
    rf.write_angles = 1;
    {
      ProfScope ps(c, "nl_repfunc_kernel");
      NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
    }
    IltArgs ia{th, ph, ts, out, N, dd, S, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale, nullptr, nullptr, c->md.time_div, 1, 0, 0, 0};
    ia.lin_wr = tab + 2 * S;
    ia.lin_wi = tab + 3 * S;
    hipError_t le;
    {
      ProfScope ps(c, "ilt_linear_stream_kernel");
      le = launch_ilt_fourier(ia, c->stream);
    }
    if (le == hipErrorInvalidValue) {
      (void)hipGetLastError();
      return fail(c, NLC_ERR_UNSUPPORTED, "nlc_model_forward: this term count does not fit the stream kernel's tiling");
    }
    NLC_HIP(c, le);
    return NLC_OK;
  }
  if (c->md.ilt.algo == NLC_ILT_DEHOOG) {
    // staged: representation function -> F_k (re, im) in HBM -> de Hoog kernel (nonlinear in F: not an MFMA epilogue)
    double* fre = pa + (N * 2 + 63) / 64 * 64;
    double* fim = fre + N * c->md.d * c->S;
    RepFuncArgs rf{};
    rf.net = c->net;
    rf.N = N;
    rf.obs = obs;
    rf.obs_stride = c->md.d;
    rf.obs_per_sample = 1;
    rf.pa = pa;
    rf.pa_stride = 2;
    rf.ts = ts;
    rf.general_t = 1;
    rf.slot = c->slot_dev;
    rf.fre = fre;
    rf.fim = fim;
    {
      ProfScope ps(c, "nl_repfunc_kernel");
      NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
    }
    IltArgs ia{nullptr, nullptr, ts, out, N, c->md.d, c->S, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale,
               fre, fim, c->md.time_div, 1, 0, 0, 0};
    ProfScope ps(c, "ilt_dehoog_kernel");
    NLC_HIP(c, launch_ilt_dehoog(ia, c->stream));
    return NLC_OK;
  }
  {
    ForwardArgs f{};
    f.net = c->net;
    f.N = N;
    f.obs = obs;
    f.pa = pa;
    f.ts = ts;
    f.out = out;
    ProfScope ps(c, "nl_forward_kernel");
    NLC_HIP(c, launch_nl_forward(f, c->stream));
  }
  return NLC_OK;
  NLC_GUARD_END(c)
}

// LaplaceRepresentationFunc.forward (w_nl.py:55-63) on explicit input rows [theta_s (S) | phi_s (S) | p (d+2)]
extern "C" int nlc_rep_func(nlc_ctx* c, const double* rep_in, int64_t N, double* theta, double* phi) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (N < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad N");
  if (N == 0) return NLC_OK;
  if (!rep_in || !theta || !phi) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  const int64_t row = 2 * (int64_t)c->S + c->P;
  RepFuncArgs rf{};
  rf.net = c->net;
  for (int i = 0; i < NLC_MAX_D; ++i) {  // the rows hold the latent p as the module sees it: no normalisation
    rf.net.state_mean[i] = 0.0;
    rf.net.state_std[i] = 1.0;
  }
  rf.N = N;
  rf.obs = rep_in + 2 * c->S;
  rf.obs_stride = row;
  rf.obs_per_sample = 1;
  rf.Kep = 1;
  rf.pa = rep_in + 2 * c->S + c->md.d;
  rf.pa_stride = row;
  rf.general_t = 1;
  rf.slot = c->slot_dev;
  rf.fre = theta;
  rf.fim = phi;
  rf.sph = rep_in;
  rf.sph_stride = row;
  rf.write_angles = 1;
  ProfScope ps(c, "nl_repfunc_kernel");
  NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}
