// Host-visible launch wrappers and argument blocks of the NLC kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nlc.h"

namespace nlc {

constexpr int kMaxTerms = 129;  // ILT terms the coefficient tables hold

// ------------------------------------------------------------------ ILT (standalone, a9)
struct IltArgs {
  const double* theta;  // (N, d, S)
  const double* phi;    // (N, d, S)
  const double* t;      // (N)
  double* x;            // (N, d)
  int64_t N;
  int d, S;
  double alpha, log_tol, scale;
  const double* fre;  // de Hoog only: when non-NULL, F_k is read from (fre, fim) (N, d, S) instead of theta/phi
  const double* fim;
  double t_div;  // de Hoog: t is divided by this (model time normalisation); 1 otherwise
  int t_stride;  // de Hoog: 1 = one t per point, 0 = t[0] for all points
  int rpp, iters;  // rows per pass / passes per block tile (set by the launcher)
  int dbg;         // 0 normal; timing experiments only: 1 memory-only, 2 arithmetic-only
  // de Hoog, planner path: when non-NULL, (fre, fim) are SLOT-major (8*nt3, N) arrays -- element e of every sample
  // contiguous, as the representation kernel's MFMA epilogue stores them -- and eidx[c*S + k] names the slot of term k
  // of dim c; one wavefront then owns 64 consecutive samples of ONE dim and every load is a full 512-B line
  const int* eidx;
  // Fourier kernel as the stream of the linear algorithms (fixed Talbot / Stehfest): when non-NULL, the (S) device tables
  // w_re, w_im replace the Fourier phase and weights and the row scale is 1/t
  const double* lin_wr;
  const double* lin_wi;
};
hipError_t launch_ilt_fourier(const IltArgs& a, hipStream_t s);
// backward of the Fourier ILT with respect to theta / phi (training through laplace_reconstruct)
struct IltBwdArgs {
  const double* theta;  // (N, d, S)
  const double* phi;    // (N, d, S)
  const double* t;      // (N)
  const double* gx;     // (N, d) upstream gradient
  double* gtheta;       // (N, d, S)
  double* gphi;         // (N, d, S)
  int64_t N;
  int d, S;
  double alpha, log_tol, scale;
  int rpp, iters;  // set by the launcher
};
hipError_t launch_ilt_fourier_bwd(const IltBwdArgs& a, hipStream_t s);
hipError_t launch_ilt_dehoog(const IltArgs& a, hipStream_t s);

struct RepInArgs {
  const double* p;  // (B, P)
  const double* t;  // (B*Tt) or (Tt)
  double* out;      // (B, Tt, 2S+P)
  int64_t B, Tt;
  int P, S, t_batched;
  double alpha, log_tol, scale;
  // linear algorithms: query points s_k = (node_re[k] + i node_im[k]) / t from device tables (NULL: Fourier / de Hoog)
  const double* node_re;
  const double* node_im;
  double t_div;  // t is divided by this before use (model time normalisation); 0 is read as 1
};
hipError_t launch_rep_inputs(const RepInArgs& a, hipStream_t s);

// fixed Talbot / Stehfest: x[n,c] = (1/t_n) sum_k (wr_k Re F_k - wi_k Im F_k), query points s_k = (node_k) / t
struct IltLinArgs {
  const double* theta;  // (N, d, S)
  const double* phi;
  const double* t;      // (N)
  double* x;            // (N, d)
  int64_t N;
  int d, S;
  const double* wr;     // (S) device tables
  const double* wi;
};
hipError_t launch_ilt_linear(const IltLinArgs& a, hipStream_t s);
// the same sum over F_k = (re, im) held SLOT-major (8*nt3, N) (staged planner path; see IltArgs::eidx)
struct IltLinSlotArgs {
  const double* fre;  // (8*nt3, N)
  const double* fim;
  const int* eidx;    // (d*S): slot of term k of dim c
  const double* t;    // one device scalar: the normalised prediction time
  double* x;          // (N, d)
  int64_t N;
  int d, S;
  const double* wr;   // (S) device tables
  const double* wi;
};
hipError_t launch_ilt_linear_slot(const IltLinSlotArgs& a, hipStream_t s);
// backward of the same with respect to theta / phi (round 3): gx (N, d) upstream gradient -> gtheta, gphi (N, d, S)
struct IltLinBwdArgs {
  const double* theta;
  const double* phi;
  const double* t;
  const double* gx;
  double* gtheta;
  double* gphi;
  int64_t N;
  int d, S;
  const double* wr;
  const double* wi;
};
hipError_t launch_ilt_linear_bwd(const IltLinBwdArgs& a, hipStream_t s);
// backward of the de Hoog ILT with respect to theta / phi (round 3; kernels_dehoog_bwd.hip): reverse mode through the QD
// table, which the kernel tapes in `scratch` (ilt_dehoog_bwd_scratch_bytes(N, d, S) bytes, owned by the launch)
struct IltDehoogBwdArgs {
  const double* theta;  // (N, d, S)
  const double* phi;
  const double* t;      // (N)
  const double* gx;     // (N, d)
  double* gtheta;       // (N, d, S)
  double* gphi;
  int64_t N;
  int d, S;
  double alpha, log_tol, scale, t_div;
  void* scratch;
};
int64_t ilt_dehoog_bwd_scratch_bytes(int64_t N, int d, int S, unsigned* grid_out);
hipError_t launch_ilt_dehoog_bwd(const IltDehoogBwdArgs& a, hipStream_t s);

// ------------------------------------------------------------------ GRU action encoder (a7)
// Action source: either an explicit window tensor (N, B, nin), or the MPPI history
// hist[k][i] = i < B-1 ? action_buffer[1+i] : u_scale * perturbed[k][i-(B-1)]  (mppi_delay.py:257-260),
// window (k,t) = hist[k][t : t+B].
struct GruArgs {
  const double* window;     // mode 0
  const double* perturbed;  // mode 1: (K, T, nu)
  const double* abuf;       // mode 1: (B, nu) device copy of action_buffer
  double u_scale;
  int mode, T;
  int t0, Tc;  // mode 1: this launch encodes horizon steps [t0, t0+Tc) of every sample (N = K*Tc)
  int64_t Kep;  // mode 1: samples per episode; sample k reads action_buffer row block k / Kep (abuf is (E, B, nu))
  int64_t N;  // windows (mode 1: K*Tc)
  int B, nin;
  int nact;  // mode 1: action dims nu; an encode_obs_time model (nin == nu + 1) gets the time channel
             // flip(arange(B)) the harness closure appends (mppi_with_model.py:110-119)
  double mean[NLC_MAX_NIN], std[NLC_MAX_NIN];
  // fragment-packed weights (device)
  // gate matrices are chunk-packed [GT chunks][KS][3 gates r,z,n][64] (see kernels_gru.hip)
  const double* Wih0p;  // KS = 1: k = input dims, k == 3 carries the folded bias
  const double* Whh0p;
  const double* Wih1p;
  const double* Whh1p;
  const double* Wop;    // [KS][1][64]   linear_out, rows 0..1
  const double* bhn0;   // (g)   b_hn layer 0
  const double* brz1;   // (2g)  b_ih + b_hh, gates r,z, layer 1
  const double* bin1;   // (g)
  const double* bhn1;   // (g)
  double bo[2];
  double* out;  // (N, 2)
  // hidden-state GEMMs on the INT8 matrix pipe (kernels_gru_i8.hip, `gru_gemm = 1`; g == 64, wave-sized tiles): the 36 gate
  // tiles of a GRU step in the order it consumes them, each block = digit fragments + recombination factors + biases
  // (nlc_pack.h: pack_gru_i8_stream)
  int use_i8;
  const signed char* i8_stream;
};
hipError_t launch_gru_encode(const GruArgs& a, int g, hipStream_t s, bool coop = false, unsigned lds_pad_bytes = 0);
hipError_t launch_gru_encode_i8(const GruArgs& a, hipStream_t s);
unsigned long long gru_i8_launch_count();  // process-wide

// ------------------------------------------------------------------ Delta-t RNN baseline (train_utils.py:589-631)
// One-layer forward GRU over the action window (hidden H) + the hidden part of linear_out: q = W_out[:, :H] h_last.
struct RnnArgs {
  const double* window;     // mode 0: (N, B, nin) raw windows
  const double* perturbed;  // mode 1: (K, T, nu)
  const double* abuf;       // mode 1: (E, B, nu)
  double u_scale;
  int mode, T;
  int64_t Kep;  // mode 1: samples per episode
  int64_t K;    // mode 1: all local samples (E * Kep)
  int64_t N;    // windows (mode 1: K * T)
  int B, nin, d;
  double mean[NLC_MAX_NIN], std[NLC_MAX_NIN];  // (0, 3) on the model's raw-input branch
  const double* Wihp;  // chunk-packed [GT][1][3][64], biases b_ih (+ b_hh for r, z) folded into column 3
  const double* Whhp;  // chunk-packed [GT][KS][3][64]
  const double* bhn;   // (H)
  const double* Wop;   // [KS][1][64]: rows 0..d-1 of linear_out.weight[:, :H]
  double* out;         // mode 0: (N, d); mode 1: (T, K, d) -- horizon-major, so the rollout reads it coalesced
};
hipError_t launch_rnn_encode(const RnnArgs& a, int hidden, hipStream_t s);

// state part of linear_out + the model's normalisation; small enough to ride in the kernel arguments
struct RnnHead {
  int d;
  double Wx[NLC_MAX_D * NLC_MAX_D];  // linear_out.weight[:, H:H+d], row-major d x d
  double wt[NLC_MAX_D];              // linear_out.weight[:, H+d]
  double b[NLC_MAX_D];               // linear_out.bias
  double mean[NLC_MAX_D], std[NLC_MAX_D];  // (0, 1) on the raw-input branch
  double time_div;                   // ts is divided by this (dt*8, or 1 on the raw branch)
};
struct RnnForwardArgs {  // DeltaTRNN.forward: out = q + Wx obs_n + wt ts_n + b
  RnnHead head;
  int64_t N;
  const double* obs;  // (N, d)
  const double* q;    // (N, d)
  const double* ts;   // (N)
  double* out;        // (N, d)
};
hipError_t launch_rnn_forward_tail(const RnnForwardArgs& a, hipStream_t s);
struct RnnRolloutArgs {  // x <- x + DeltaTRNN(x, window_t, ts_pred), running cost, perturbation cost
  RnnHead head;
  int64_t K, Kep;
  int T, nu, env;  // env: running cost, -1 = none (cost_external)
  int state_per_sample;
  const double* state0;
  const double* q;  // (T, K, d)
  const double* perturbed;
  const double* noise;
  const double* U;
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale, ts;
  int noise_abs_cost;
  double* states;
  double* cost_total;
};
hipError_t launch_rnn_rollout(const RnnRolloutArgs& a, hipStream_t s);

// ------------------------------------------------------------------ representation MLP + ILT + rollout
struct NlNetArgs {
  int d, S, h, nt3;      // nt3 = layer-3 output tiles (theta/phi interleaved slot layout)
  int n_even_groups;     // ILT groups (4 slots each) below this index hold even-k terms (cos), the rest odd-k (sin)
  const double* W1p;     // [2][HT][64]        latent part of layer 1 (P <= 8 inputs)
  const double* W1s;     // [KSS][HT][64]      sphere-coordinate part (general-t mode only)
  const double* b1;      // (h)                ROLLOUT: bias with the constant sphere inputs folded in
  const double* W2p;     // [h/4][HT][64]
  const double* b2;      // (h)
  const double* W3p;     // [h/4][nt3][64]     rows permuted into the slot layout
  const double* b3p;     // (16*nt3)           same permutation
  const double* Cp;      // [2*nt3][64]        ILT coefficient matrix fragments (rows = dims)
  // fixed Talbot / Stehfest models on the fused rollout (round 3, LIN instances of the rollout kernels): the reconstruction
  // x = sum_k (w_re,k / t) Re F_k - (w_im,k / t) Im F_k is TWO epilogue MFMAs per slot group -- Cp holds w_re / t against
  // R cos(theta), Cp2 holds -w_im / t against R sin(theta) (both folded for the planner's constant t); lin = 1 selects them
  const double* Cp2;
  int lin;
  double state_mean[NLC_MAX_D], state_std[NLC_MAX_D];
  double alpha, log_tol, scale, time_div;
};

// Episodes (nlc_mppi_desc.E): K counts ALL local samples (E * Kep); sample k belongs to episode e = k / Kep and
// reads that episode's state0 row, action_buffer block and U block.  E == 1 <=> Kep == K.
struct RolloutArgs {
  NlNetArgs net;
  int64_t K, Kep;
  int T, nu, B, env;
  int state_per_sample;
  const double* state0;     // (E,d) or (K,d) device
  const double* pa;         // (K, T, 2) GRU latents
  const double* perturbed;  // (K, T, nu)
  const double* noise;      // (K, T, nu) bounded noise
  const double* U;          // (E, T, nu)
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale;
  int noise_abs_cost;
  double tn;                // normalised prediction time (constant over the rollout)
  // horizon chunking: this launch runs steps [t_begin, t_end); state and the two cost sums are carried between
  // launches in xcarry (K, d) / ccarry (K, 2) so the GRU encode of later steps can overlap earlier rollout steps
  int t_begin, t_end;
  double* xcarry;
  double* ccarry;
  double* states;           // (K, T, d) or NULL
  double* cost_total;       // (K)
};
hipError_t launch_nl_rollout(const RolloutArgs& a, hipStream_t s, int force_variant = 0);

struct ForwardArgs {
  NlNetArgs net;
  int64_t N;
  const double* obs;  // (N, d)
  const double* pa;   // (N, 2)
  const double* ts;   // (N) raw ts_pred
  double* out;        // (N, d)
  int const_t;        // 1: one query time for every row -- tn below, net.b1 = the folded bias, ts unused
  double tn;          // ts_pred / time_div of the constant-time form
};
hipError_t launch_nl_forward(const ForwardArgs& a, hipStream_t s);

// x <- x + dx, running cost, state store: the per-step tail of the staged (de Hoog) planner path
struct StepTailArgs {
  int64_t K, Kep;
  int T, t, nu, d, env, first, last;
  int state_per_sample;
  const double* state0;  // read when first
  double* x;             // (K, d) carried state
  const double* dx;      // (K, d)
  double* ccarry;        // (K, 2) running cost / perturbation cost
  const double* perturbed;
  const double* noise;
  const double* U;
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale;
  int noise_abs_cost;
  double* states;      // (K, T, d) or NULL
  double* cost_total;  // (K), written when last
};
hipError_t launch_step_tail(const StepTailArgs& a, hipStream_t s);

// representation function only: F_k (re, im) of every Laplace term -> (N, d, S) arrays (de Hoog path)
struct RepFuncArgs {
  NlNetArgs net;
  int64_t N;
  const double* obs;  // (N, d), or (E, d) rows broadcast over the Kep samples of each episode
  int64_t obs_stride; // doubles between obs rows (d for a dense tensor)
  int obs_per_sample;
  int64_t Kep;
  const double* pa;   // GRU latents, row n at pa + n*pa_stride
  int64_t pa_stride;
  const double* ts;   // (N) raw ts_pred (general_t)
  double tn;          // constant normalised time (!general_t; net.b1 must be the folded bias)
  int general_t;
  const int* slot;    // (8*nt3) slot -> c*S + k or -1
  double* fre;        // (N, d, S)
  double* fim;
  // LaplaceRepresentationFunc.forward on explicit input rows (general_t only): sphere inputs [theta_s | phi_s] of row n at
  // sph + n * sph_stride (NULL: computed from ts); write_angles: store the module's (theta, phi) instead of F (re, im)
  const double* sph;
  int64_t sph_stride;
  int write_angles;
  int slot_major;     // 1: fre / fim are (8*nt3, N) slot-major (planner path; see IltArgs::eidx)
  // planner path, horizon step t > 0: the tail of step t-1 runs as this launch's prologue -- x <- xcarry + dx, store,
  // running + perturbation cost into ccarry (StepTailArgs semantics) -- and the new x is the observation
  int tail_prev;      // 0: no prologue (obs is read as given)
  StepTailArgs tail;  // tail.t = the PREVIOUS step
  int split;          // planner path: 1 = one workgroup per 16-sample tile (nl_repfunc_split_kernel), 0 = one wave per tile
};
hipError_t launch_nl_repfunc(const RepFuncArgs& a, hipStream_t s);


// the whole step chain of the de Hoog planner as one persistent launch (kernels_dehoog_chain.hip): a workgroup owns 64
// consecutive samples for all T horizon steps.  Single planner (E == 1), hidden_units 128, 17 or 33 terms.
struct DehoogChainArgs {
  NlNetArgs net;            // net.b1: the bias with the constant sphere inputs folded in
  int64_t K;
  int T, nu, env;
  int state_per_sample;
  const double* state0;     // (d) or (K, d)
  const double* pa;         // (K, T, 2) GRU latents of the hoisted encode
  const double* perturbed;  // (K, T, nu)
  const double* noise;
  const double* U;          // (T, nu)
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale;
  int noise_abs_cost;
  double tn;                // normalised prediction time
  const int* slot;          // (8 nt3) layer-3 slot -> c*S + k, -1 = padding
  const int* eidx;          // (d*S)   term k of dim c -> slot
  double* fre;              // one private (8 nt3) x (16 block_tiles) block per workgroup's samples: ceil(K / 64) * 64 * 8 nt3 doubles
  double* fim;
  double* states;           // (K, T, d) or NULL
  double* cost_total;       // (K)
  int phases;               // 3; tools only: 1 = the representation phase alone, 2 = the QD phase alone (timing breakdown)
};
hipError_t launch_nl_dehoog_chain(const DehoogChainArgs& a, int block_tiles, hipStream_t s);
bool nl_dehoog_chain_available(int h, int nt3, int S);

// ------------------------------------------------------------------ oracle-dynamics rollout (§8f-1)
struct OracleRolloutArgs {
  int64_t K, Kep;
  int T, nu, B, d, env, delay, friction;
  int cost_env;  // env of the running cost, -1 = none (the caller adds its own cost)
  int state_per_sample;
  const double* state0;
  const double* abuf;  // (E, B, nu)
  const double* perturbed;
  const double* noise;
  const double* U;
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale, ts;
  int noise_abs_cost;
  double* states;
  double* cost_total;
};
hipError_t launch_oracle_rollout(const OracleRolloutArgs& a, hipStream_t s);

// ------------------------------------------------------------------ NODE baseline (train_utils.py:637-738)
struct NodeNetArgs {
  int d, aug, nu;      // state_dim, augment_dim, action_dim: ODE-function input [x (d) | aug | u (nu)], <= 12 entries
  const double* W1p;   // [3][HT][64]     inputs padded to 12
  const double* b1;    // (16*HT)         padded with zeros (tanh(0) = 0: padded hidden units contribute nothing)
  const double* W2p;   // [4*HT][HT][64]
  const double* b2;    // (16*HT)
  const double* W3p;   // [4*HT][1][64]   rows 0 .. d+aug-1
  const double* b3;    // (16)
  double state_mean[NLC_MAX_D], state_std[NLC_MAX_D];  // (0, 1) if !normalize
  int nsub;            // Euler sub-steps of the fixed grid over [0, ts_pred / time_div] (step_size 0.05)
  double hsub[8];
};
struct NodeRolloutArgs {
  NodeNetArgs net;
  int64_t K, Kep;
  int T, nu, env;
  int state_per_sample;
  const double* state0;
  const double* perturbed;
  const double* noise;
  const double* U;
  double sigma_inv[NLC_MAX_NU * NLC_MAX_NU];
  double lambda_, u_scale;
  int noise_abs_cost;
  double* states;
  double* cost_total;
};
struct NodeForwardArgs {
  NodeNetArgs net;
  int64_t N;
  const double* obs;     // (N, d)
  const double* action;  // (N, nu): window[:, -1, :]
  double* out;           // (N, d)
};
hipError_t launch_node_rollout(const NodeRolloutArgs& a, int ht, hipStream_t s);
hipError_t launch_node_forward(const NodeForwardArgs& a, int ht, hipStream_t s);

// ------------------------------------------------------------------ env side of the evaluation loop
// step_env (mppi_with_model.py:193-216) for E independent envs: get_action (delay buffer, :25-28), one Euler step of
// the env's torch_rhs on the reduced state (base_env.py:136-173 with solver="euler", ts = [0, dt]), get_obs, reward.
struct EnvStepArgs {
  int env, friction, B, nu, delay;
  int64_t E;
  double dt;
  double* state;         // (E, n) reduced state (angles), in/out; NULL action = observation only
  double* abuf;          // (E, B, nu) in/out: rolled by one row, the new action appended
  const double* action;  // (E, nu) the planner's (un-delayed) action
  double* obs;           // (E, d) out: trig observation of the new state
  double* reward;        // (E) out: diff_reward(new state, applied action); may be NULL
};
hipError_t launch_env_step(const EnvStepArgs& a, hipStream_t s);

// ------------------------------------------------------------------ MPPI sampling / weighting
constexpr int kMaxInlineAbuf = 32;  // action_buffer doubles carried in the kernel arguments (B*nu <= 32)
struct PerturbArgs {
  int64_t K, Kep, K_global, k_offset;  // K_global / k_offset are per episode
  int T, nu, E;
  const double* U_old;  // (E, T, nu) before the shift
  double* U_new;        // (E, T, nu) after roll(-1) + u_init
  double* noise;        // in (rng==0) / out
  double* perturbed;
  double* actions;      // may be NULL
  double u_scale;
  int has_bounds, sample_null_action, rng;
  double u_min[NLC_MAX_NU], u_max[NLC_MAX_NU], u_init[NLC_MAX_NU], mu[NLC_MAX_NU];
  double chol[NLC_MAX_NU * NLC_MAX_NU];
  uint64_t seed, counter;
  // single planner: the command's state (d) and action_buffer (B, nu) ride in the kernel arguments and the shift
  // kernel stores them into the workspace -- no host-to-device copy commands on the command's critical path
  int n_state_in, n_abuf_in;  // 0 = not carried (batched / per-sample state / oversize: copied by the host API)
  double* state_dst;
  double* abuf_dst;
  double state_in[NLC_MAX_D];
  double abuf_in[kMaxInlineAbuf];
  // words the perturb kernel zeroes for a later kernel of the same command (the fused planner's sync block)
  unsigned* zero_words;
  int64_t n_zero_words;
  int fused_shift;  // 1: the perturb kernel also does shift_U_kernel's work (U <- roll(U, -1), the staged inputs)
};
hipError_t launch_shift_U(const PerturbArgs& a, hipStream_t s);
hipError_t launch_perturb(const PerturbArgs& a, hipStream_t s);

struct WeightArgs {
  int64_t Kep;  // samples per episode; blockIdx.y = episode
  int T, nu, E;
  double lambda_;
  const double* cost;   // (E, Kep)
  const double* noise;  // (E, Kep, T, nu)
  double* tile_part;    // (E, nblk, 2 + T*nu): (beta_b, eta_b, S_b) of every 16-sample tile
  double* chunk_part;   // (E, ceil(nblk/64), 2 + T*nu): 64-tile chunk sums (large populations: launch_weights)
  double* partials;     // (E, 2 + T*nu)
  int nblk;             // weight tiles per episode
  // non-NULL after a fused launch that did NOT fold the weights itself (batched episodes, fused_inline without bit 1, cost
  // callables): the sync block's kFusedTimeout word -- non-zero = a workgroup of that launch gave up, and the workgroup that
  // stores eta_r marks every episode's partial row (kPartialInvalidEta) exactly as the in-launch fold does (fused_weight_rank)
  const unsigned* gave_up;
};
// partial rows are (beta_r, eta_r, S_r[T*nu]); eta_r = kPartialInvalidEta marks a shard whose rollout launch gave up (the
// fused planner body's bounded waits): merge_kernel on every rank then leaves U alone and reports it (MergeArgs::status_pinned)
constexpr double kPartialInvalidEta = -1.0;
constexpr int kWeightTile = 16;  // = the rollout kernels' MFMA tile: a tile's samples become final together
inline int weight_tiles(int64_t Kep) { return (int)((Kep + kWeightTile - 1) / kWeightTile); }
hipError_t launch_weights(const WeightArgs& a, hipStream_t s);

struct MergeArgs {
  int64_t Kep;  // samples per episode; blockIdx.y = episode
  int T, nu, G, rank, u_per_command, E;
  double lambda_, u_scale;
  const double* gathered;  // (G, E, 2 + T*nu)
  double* U;               // (E, T, nu) updated in place
  double* cost_nz;         // (E, Kep) out
  double* omega;           // (E, Kep) or NULL
  double* action;          // (E, u_per_command * nu) device
  double* action_pinned;   // same, in pinned host memory the kernel stores to directly (NULL = not wanted)
  double* beta_eta;        // (E, 2) device: merged beta, eta
  const double* cost;      // (E, Kep) total costs: cost_nz = exp(-(cost - beta)/lambda) with the merged beta
  unsigned* zero_words;    // words zeroed for the NEXT command (the fused planner body's sync block); may be NULL
  int64_t n_zero_words;
  // single planner, host action wanted: after action_pinned the kernel stores `seq` here (pinned, system scope) -- the host
  // spins on this word instead of sleeping in hipStreamSynchronize (an interrupt wake-up costs 10-20 us per command)
  unsigned long long* seq_pinned;
  unsigned long long seq;
  // pinned word the kernel sets to 1 -- before seq -- when some rank's partial row is marked invalid (kPartialInvalidEta): U, the
  // action (NaN), cost_nz and omega are then NOT produced; the host clears the word when it has seen it.  May be NULL.
  unsigned* status_pinned;
};
hipError_t launch_merge(const MergeArgs& a, hipStream_t s);

// One-launch planner body for small populations (kernels_fused.hip): GRU encode + split rollout as roles of one grid,
// and -- for the single planner with device noise -- the command's sampling / bounding (in the encoder role) and its
// importance-weight reduction (after the last rollout tile) as further phases of the same launch.
// `sync` is a block of unsigned words that is zero when the launch starts (zeroed by the previous command's merge kernel,
// by the command's own perturb kernel when that still runs, or by a memset):
constexpr int kFusedEncTicket = 0, kFusedRollTicket = 1, kFusedTimeout = 2, kFusedCensusTicket = 3;
// progress counters (diagnostics; one relaxed atomic add each): workgroups entered / rollout tiles started / finished /
// encoder tiles published / workgroups exited
constexpr int kFusedStatEntered = 4, kFusedStatRollStart = 5, kFusedStatRollDone = 6, kFusedStatEncDone = 7, kFusedStatExited = 8;
// timeline (low 32 bits of the 100 MHz s_memrealtime counter; "first" values are stored complemented so that the zeroed
// word works with atomic max): first entry, first / last rollout past its first hand-off, first / last rollout done,
// last encoder tile published
constexpr int kFusedTimeEntry = 9, kFusedTimeRollBeginFirst = 10, kFusedTimeRollBeginLast = 11, kFusedTimeRollEndFirst = 12,
              kFusedTimeRollEndLast = 13, kFusedTimeEncLast = 14;
constexpr int kFusedCuOcc = 16;              // 2048 per-CU arrival counters (XCC_ID << 8 | SE/SH/CU id)
constexpr int kFusedCuState = 16 + 2048;     // 2048 per-CU words: 1 the CU's first workgroup walks a chain, 2 it has finished
// in-launch weight reduction: rollout tiles whose weight-tile partial has been published
constexpr int kFusedCostDone = 16 + 4096;
constexpr int kFusedFlags = 16 + 4096 + 16;  // (T, ntk) one word per encoder tile, then one (ntk) row of rollout-tile owner
                                             // tickets (the first add owns the tile)
// (+ (T, ntk) chain-step time stamps, written by the trace build only)
inline size_t fused_sync_words(int T, int64_t K) { return (size_t)kFusedFlags + (size_t)(2 * T + 1) * (size_t)((K + 15) / 16); }
constexpr int kFusedMaxInlineB = 8;  // action-buffer rows the in-launch sampling stages in LDS
struct FusedCtl {   // role assignment; passed to the kernel by value
  unsigned* sync;
  unsigned* timeout_host;  // pinned host word: non-zero = a rollout workgroup gave up waiting (command lost)
  int ntk;         // 16-sample tiles per horizon step
  int n_enc;       // T * ntk encoder tiles
  int chain_first_tiles;  // encoder tiles every wave of a chain's workgroup encodes before the chain starts
  int partner_tiles;      // >= 0: the other workgroup of a chain's CU sleeps after this many tiles per wave until the chain
                          // is done; < 0: it never sleeps
  int roll_cap;    // rollout workgroups that start right away (one per CU); the other tiles drain after the encoders
  int adaptive_q8; // > 0: partners sleep / help by the feedback rule of fused_encode; value = 256 * (tile time / chain step time)
  int pool_wgs;    // workgroups on CUs without a chain (they encode throughout)
  int inline_perturb;  // 1: no perturb kernel ran -- encoder tile (t, j) samples / bounds the actions of its windows
                       // (FusedArgs::p, device Philox) and publishes perturbed / noise / actions of step t with its latents;
                       // the command's state and action_buffer are read from the kernel arguments (p.state_in / p.abuf_in)
  int inline_weights;  // 1: the importance-weight reduction (FusedArgs::w) runs after the last rollout tile
  unsigned spin_limit; // polls before a waiting wave gives up (the command is then lost and reported)
  int test_drop_tile;  // tests only: the encoder tile with this ticket is never published (-1: none)
};
struct FusedArgs {   // the kernel's one by-value argument
  RolloutArgs r;   // r.pa: the (T, K, 2) HORIZON-major latent tensor (written and read inside the launch)
  GruArgs g;       // mode 1 fields filled in; g.out unused
  PerturbArgs p;   // inline_perturb
  WeightArgs w;    // inline_weights
  FusedCtl ctl;
};
// bpc_built: the instance compiled for 3 or 4 workgroups per CU (kernels_fused.hip)
hipError_t launch_nl_plan_fused(const FusedArgs& a, int g, unsigned grid, int bpc_built, hipStream_t s);
hipError_t fused_max_resident_blocks(int h, int bpc_built, int* blocks_per_cu);


}  // namespace nlc
