// Host side of libnlc_hip.so, baselines unit: the env side of the evaluation loop (nlc_env_step / nlc_env_obs) and the
// Delta-t RNN / NODE baseline dynamics models (upload + forward).
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

// =================================================================================== env side of the loop
extern "C" int nlc_env_step(nlc_ctx* c, int env, int friction, double dt, int delay, int64_t E, int B, int nu,
                            double* state, double* action_buffer, const double* action, double* obs, double* reward) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  static const int env_nu[3] = {1, 1, 2};
  if (env < 0 || env > 2) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  if (nu != env_nu[env]) return fail(c, NLC_ERR_BAD_SHAPE, "nu does not match the env's action space");
  if (E < 0 || B < 1 || delay < 0 || delay > B - 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad E / B / delay");
  if (E == 0) return NLC_OK;
  if (!state || !action_buffer || !action || !obs) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  EnvStepArgs a{env, friction, B, nu, delay, E, dt, state, action_buffer, action, obs, reward};
  ProfScope ps(c, "env_step_kernel");
  NLC_HIP(c, launch_env_step(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_env_obs(nlc_ctx* c, int env, int64_t E, const double* state, double* obs) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (env < 0 || env > 2) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  if (E < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad E");
  if (E == 0) return NLC_OK;
  if (!state || !obs) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  EnvStepArgs a{env, 0, 1, 1, 0, E, 0.0, const_cast<double*>(state), nullptr, nullptr, obs, nullptr};
  ProfScope ps(c, "env_step_kernel");
  NLC_HIP(c, launch_env_step(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== Delta-t RNN baseline
static int64_t rnn_blob_size(const nlc_rnn_desc* d) {
  const int64_t H = d->hidden;
  return 3 * H * d->nin + 3 * H * H + 6 * H + (int64_t)d->d * (H + d->d + (d->time_input ? 1 : 0)) + d->d;
}
extern "C" int64_t nlc_rnn_blob_size(const nlc_rnn_desc* d) { return d ? rnn_blob_size(d) : -1; }

extern "C" int nlc_set_rnn_model(nlc_ctx* c, const nlc_rnn_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (d->hidden != 64 && d->hidden != 128 && d->hidden != 160)
    return fail(c, NLC_ERR_UNSUPPORTED, "DeltaTRNN hidden_units must be 64, 128 or 160");
  if (d->d < 1 || d->d > NLC_MAX_D) return fail(c, NLC_ERR_UNSUPPORTED, "state_dim must be in 1..8");
  if (d->nin < 1 || d->nin > NLC_MAX_NIN) return fail(c, NLC_ERR_UNSUPPORTED, "GRU input dim must be in 1..3");
  if (!(d->time_div != 0.0)) return fail(c, NLC_ERR_BAD_ARG, "time_div must be non-zero");
  if (n != rnn_blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int H = d->hidden, dd = d->d, nin = d->nin, F = H + dd + (d->time_input ? 1 : 0);
  Blob b{w, n};
  const double* Wih = b.take((int64_t)3 * H * nin);
  const double* Whh = b.take((int64_t)3 * H * H);
  const double* bih = b.take(3 * H);
  const double* bhh = b.take(3 * H);
  const double* Wo = b.take((int64_t)dd * F);
  const double* bo = b.take(dd);
  DeviceArena ar;
  // input weights with the biases folded into input column 3 (x = [a_0..a_{nin-1}, 0.., 1]), as for the NL encoder
  std::vector<double> Wihb((size_t)3 * H * 4, 0.0);
  for (int r = 0; r < 3 * H; ++r) {
    for (int j = 0; j < nin; ++j) Wihb[(size_t)r * 4 + j] = Wih[(size_t)r * nin + j];
    Wihb[(size_t)r * 4 + 3] = bih[r] + (r < 2 * H ? bhh[r] : 0.0);
  }
  const size_t o_Wih = ar.push(pack_gru_chunked(Wihb.data(), 4, 4, H));
  const size_t o_Whh = ar.push(pack_gru_chunked(Whh, H, H, H));
  const size_t o_bhn = ar.push(std::vector<double>(bhh + 2 * H, bhh + 3 * H));
  const size_t o_Wo = ar.push(pack_A(Wo, F, H, identity_rows(dd)));
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->rnn_base) hipFree(c->rnn_base);
  c->rnn_base = base;
  c->rd = *d;
  RnnArgs& R = c->rnn;
  R = RnnArgs{};
  R.nin = nin;
  R.d = dd;
  for (int j = 0; j < nin; ++j) {
    R.mean[j] = d->action_mean[j];
    R.std[j] = d->action_std[j];
  }
  R.Wihp = base + o_Wih;
  R.Whhp = base + o_Whh;
  R.bhn = base + o_bhn;
  R.Wop = base + o_Wo;
  RnnHead& Hd = c->rnn_head;
  Hd = RnnHead{};
  Hd.d = dd;
  for (int i = 0; i < dd; ++i) {
    for (int j = 0; j < dd; ++j) Hd.Wx[i * dd + j] = Wo[(size_t)i * F + H + j];
    Hd.wt[i] = d->time_input ? Wo[(size_t)i * F + H + dd] : 0.0;
    Hd.b[i] = bo[i];
    Hd.mean[i] = d->state_mean[i];
    Hd.std[i] = d->state_std[i];
  }
  Hd.time_div = d->time_div;
  c->has_rnn = true;
  if (c->has_mppi && c->pd.dynamics == NLC_DYN_DTRNN) c->has_mppi = false;  // re-configure against the new weights
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_rnn_forward(nlc_ctx* c, const double* obs, const double* window, const double* ts, int64_t N, int B,
                               double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_rnn) return fail(c, NLC_ERR_STATE, "nlc_set_rnn_model has not been called");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !ts || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  RnnArgs a = c->rnn;
  a.mode = 0;
  a.window = window;
  a.N = N;
  a.B = B;
  a.out = (double*)ws;
  {
    ProfScope ps(c, "rnn_encode_kernel");
    NLC_HIP(c, launch_rnn_encode(a, c->rd.hidden, c->stream));
  }
  RnnForwardArgs f{};
  f.head = c->rnn_head;
  f.N = N;
  f.obs = obs;
  f.q = (const double*)ws;
  f.ts = ts;
  f.out = out;
  ProfScope ps(c, "rnn_forward_tail_kernel");
  NLC_HIP(c, launch_rnn_forward_tail(f, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== NODE baseline
namespace {
int64_t node_blob_size(const nlc_node_desc* d) {
  const int64_t H = d->hidden, dy = d->d + d->augment_dim, in = dy + d->nu;
  return H * in + H + H * H + H + dy * H + dy;
}
}  // namespace
namespace nlc {
namespace host {
// step sizes of torchdiffeq's fixed-grid solver over [0, t_end] (restated: oracle/node_model.py::euler_substeps)
int node_substeps(double t_end, double step, double* h, int max_n) {
  const int niters = (int)std::ceil(t_end / step + 1.0);
  if (niters < 2 || niters - 1 > max_n) return -1;
  double prev = 0.0;
  for (int i = 1; i < niters; ++i) {
    const double tk = (i == niters - 1) ? t_end : (double)i * step;
    h[i - 1] = tk - prev;
    prev = tk;
  }
  return niters - 1;
}
}  // namespace host
}  // namespace nlc
extern "C" int64_t nlc_node_blob_size(const nlc_node_desc* d) { return d ? node_blob_size(d) : -1; }

extern "C" int nlc_set_node_model(nlc_ctx* c, const nlc_node_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (d->hidden < 1 || d->hidden > 272) return fail(c, NLC_ERR_UNSUPPORTED, "NODE hidden_units must be in 1..272");
  if (d->d < 1 || d->augment_dim < 0 || d->d + d->augment_dim > 8 || d->d > NLC_MAX_D)
    return fail(c, NLC_ERR_UNSUPPORTED, "state_dim + augment_dim must be <= 8");
  if (d->nu < 1 || d->nu > NLC_MAX_NU) return fail(c, NLC_ERR_UNSUPPORTED, "nu must be 1 or 2");
  if (!(d->time_div != 0.0) || !(d->step_size > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "bad time_div / step_size");
  if (n != node_blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int H = d->hidden, dy = d->d + d->augment_dim, in = dy + d->nu;
  const int ht = H <= 64 ? 4 : (H <= 128 ? 8 : 17), Hp = 16 * ht;
  Blob b{w, n};
  const double* W1 = b.take((int64_t)H * in);
  const double* b1 = b.take(H);
  const double* W2 = b.take((int64_t)H * H);
  const double* b2 = b.take(H);
  const double* W3 = b.take((int64_t)dy * H);
  const double* b3 = b.take(dy);
  // pad the hidden width to Hp rows / columns with zeros; inputs to 12 columns
  std::vector<double> W1z((size_t)Hp * 12, 0.0), W2z((size_t)Hp * Hp, 0.0), W3z((size_t)16 * Hp, 0.0), b1z(Hp, 0.0),
      b2z(Hp, 0.0), b3z(16, 0.0);
  for (int r = 0; r < H; ++r) {
    for (int j = 0; j < in; ++j) W1z[(size_t)r * 12 + j] = W1[(size_t)r * in + j];
    for (int j = 0; j < H; ++j) W2z[(size_t)r * Hp + j] = W2[(size_t)r * H + j];
    b1z[r] = b1[r];
    b2z[r] = b2[r];
  }
  for (int r = 0; r < dy; ++r) {
    for (int j = 0; j < H; ++j) W3z[(size_t)r * Hp + j] = W3[(size_t)r * H + j];
    b3z[r] = b3[r];
  }
  DeviceArena ar;
  const size_t o_W1 = ar.push(pack_A(W1z.data(), 12, 12, identity_rows(Hp)));
  const size_t o_b1 = ar.push(b1z);
  const size_t o_W2 = ar.push(pack_A(W2z.data(), Hp, Hp, identity_rows(Hp)));
  const size_t o_b2 = ar.push(b2z);
  const size_t o_W3 = ar.push(pack_A(W3z.data(), Hp, Hp, identity_rows(16)));
  const size_t o_b3 = ar.push(b3z);
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->node_base) hipFree(c->node_base);
  c->node_base = base;
  c->nd = *d;
  c->node_ht = ht;
  NodeNetArgs& N = c->node;
  N = NodeNetArgs{};
  N.d = d->d;
  N.aug = d->augment_dim;
  N.nu = d->nu;
  N.W1p = base + o_W1;
  N.b1 = base + o_b1;
  N.W2p = base + o_W2;
  N.b2 = base + o_b2;
  N.W3p = base + o_W3;
  N.b3 = base + o_b3;
  for (int i = 0; i < NLC_MAX_D; ++i) {
    N.state_mean[i] = i < d->d ? d->state_mean[i] : 0.0;
    N.state_std[i] = i < d->d ? d->state_std[i] : 1.0;
  }
  c->has_node = true;
  if (c->has_mppi && c->pd.dynamics == NLC_DYN_NODE) c->has_mppi = false;
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_node_forward(nlc_ctx* c, const double* obs, const double* action, double ts_pred, int64_t N,
                                double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_node) return fail(c, NLC_ERR_STATE, "nlc_set_node_model has not been called");
  if (N < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad N");
  if (N == 0) return NLC_OK;
  if (!obs || !action || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  NodeForwardArgs f{};
  f.net = c->node;
  f.net.nsub = node_substeps(ts_pred / c->nd.time_div, c->nd.step_size, f.net.hsub, 8);
  if (f.net.nsub < 0) return fail(c, NLC_ERR_UNSUPPORTED, "prediction time needs more than 8 Euler sub-steps (or is <= 0)");
  f.N = N;
  f.obs = obs;
  f.action = action;
  f.out = out;
  ProfScope ps(c, "node_forward_kernel");
  NLC_HIP(c, launch_node_forward(f, c->node_ht, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}
