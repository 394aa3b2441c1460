/*
 * nlc.h -- C ABI of libnlc_hip.so: the MI355X (gfx950) planning hot path of Neural Laplace Control.
 *
 * The reference (samholt/NeuralLaplaceControl) is pure Python and has NO FFI layer; its boundary for
 * this path is three Python call sites.  Each entry point below names the reference interface it
 * replaces (file:line, relative to the reference tree).  The Python mirror of those interfaces
 * (neurallaplacecontrol_amd/) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns an int status (NLC_OK = 0, negative = error); text via nlc_last_error().
 *     No C++ exception crosses this boundary.
 *   - "_dev" pointers are DEVICE pointers (HBM of the ctx's GPU; e.g. torch.Tensor.data_ptr()),
 *     "_host" pointers are host pointers.  All payloads are float64, row-major, contiguous.
 *   - the caller owns every pointer for the duration of the call only; the ctx copies what it keeps.
 *   - kernels are enqueued on the stream given to nlc_set_stream() (until then: a stream the ctx created).
 *     Functions that return host data synchronise that stream before returning; the others are
 *     asynchronous with respect to the host, ordered on the stream.
 *   - a ctx is not thread-safe: one ctx per (process, device).  HIP is initialised lazily inside
 *     nlc_create(), never at library load (spawn-safe, cf. run_exp_multi.py:145,207).
 */
#ifndef NLC_H_
#define NLC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLC_ABI_VERSION 9

#define NLC_OK 0
#define NLC_ERR_BAD_ARG (-1)
#define NLC_ERR_BAD_SHAPE (-2)
#define NLC_ERR_HIP (-3)
#define NLC_ERR_UNSUPPORTED (-4)
#define NLC_ERR_STATE (-5)
#define NLC_ERR_COMM (-6) /* RCCL reported an error (message in nlc_last_error) */
/* nlc_mppi_finish only, caller-owned collective (gathered_dev != NULL, G > 1): the command was re-run on the two-launch body on
 * every rank; buf->partials hold the new rows -- all-gather them again and call nlc_mppi_finish again.  Not an error. */
#define NLC_AGAIN 1

#define NLC_MAX_NU 2   /* action dims (cartpole/pendulum 1, acrobot 2) */
#define NLC_MAX_NIN 3  /* GRU input dims = nu + encode_obs_time */
#define NLC_MAX_D 8    /* observation dims (3..6 in the reference envs) */

/* ILT algorithms (torchlaplace `ilt_algorithm`, call site w_nl.py:141) */
#define NLC_ILT_FOURIER 0
#define NLC_ILT_DEHOOG 1
/* the closed-form LINEAR algorithms torchlaplace also offers (reference knob config.py:36 nl_ilt_algorithm): stand-alone
 * reconstruction (nlc_ilt_rep_inputs / nlc_ilt_reconstruct / its backward), nlc_model_forward, and the planner -- on LIN
 * instances of the rollout kernels by default, on the staged all-HIP path with option "linear_fused" = 0 (see nlc_set_option) */
#define NLC_ILT_FIXED_TALBOT 2 /* "fixed_tablot": Abate & Valko, M = terms nodes, r = 2M/5 (mpmath FixedTalbot) */
#define NLC_ILT_STEHFEST 3     /* "stehfest": Gaver-Stehfest, terms even (mpmath Stehfest) */

/* env running costs (mppi_with_model.py:145-171 -> ctcartpole.py:289-346, ctpendulum.py:139-155,
 * ctacrobot.py:233-255) and oracle dynamics (oracle.py:11-224) */
#define NLC_ENV_CARTPOLE 0
#define NLC_ENV_PENDULUM 1
#define NLC_ENV_ACROBOT 2
/* cartpole WITHOUT the trig observation: CTCartpole(obs_trans=False), nx = 4, state [x, xdot, theta, thetadot]
 * (ctcartpole.py:60; reward branch :297-300; oracle dynamics branch oracle.py:38-44, 80-86) -- BASELINE's literal
 * "state_dim=4".  Planner only (running cost, oracle dynamics, NL dynamics); nlc_env_step keeps the three harness envs. */
#define NLC_ENV_CARTPOLE_NOTRIG 3

/* rollout dynamics */
#define NLC_DYN_NL 0     /* state + NeuralLaplaceModel(state, window, ts_pred)  (mppi_with_model.py:103-122) */
#define NLC_DYN_ORACLE 1 /* oracle.{cartpole,pendulum,acrobot}_dynamics_dt_delay (mppi_with_model.py:129-143) */
#define NLC_DYN_EXTERNAL 2 /* caller-supplied dynamics/cost callables: the caller runs the horizon loop
                              (mppi_delay.py:271-296) itself between nlc_mppi_rollout and nlc_mppi_weights */
#define NLC_DYN_DTRNN 3    /* state + DeltaTRNN(state, window, ts_pred): the reference's Delta-t RNN baseline
                              (train_utils.py:589-631) behind the same harness closure (mppi_with_model.py:103-122) */
#define NLC_DYN_NODE 4     /* state + NODE(state, window, ts_pred): the neural-ODE baseline (train_utils.py:664-724) */

typedef struct nlc_ctx nlc_ctx;

/* ---- context -------------------------------------------------------------------------------- */
int nlc_abi_version(void);
int nlc_create(int device, nlc_ctx** out);
void nlc_destroy(nlc_ctx* ctx);
const char* nlc_last_error(const nlc_ctx* ctx); /* ctx may be NULL: error of the last failed nlc_create */
/* hipStream_t to enqueue on from now on, used as given: NULL is HIP's legacy default (null) stream, which
 * is what torch.cuda.current_stream().cuda_stream reports for torch's default stream. */
int nlc_set_stream(nlc_ctx* ctx, void* hip_stream);
int nlc_synchronize(nlc_ctx* ctx);
/* Planner tuning knobs (none changes a result beyond last-bit rounding of the reductions they do not touch; they pick
 * which hand-written rollout body a command of NLC_DYN_NL / Fourier runs):
 *   "rollout_variant"    0 auto (default) | 1 one wavefront per 16-sample tile | 2 latency-split, one workgroup per tile
 *                        | 3 fused one-launch body (GRU encode + split rollout as roles of one persistent grid)
 *   "fused_max_samples"  auto picks the fused body up to this many local samples (default 4096: one chain per CU; one
 *                        GPU's shard of BASELINE configs[1] at 4 and 8 GPUs)
 *   "fused_roll_cap"     rollout workgroups the fused body starts right away, one per CU (0 = auto: one per 16-sample
 *                        tile, at most one per CU)
 *   "repfunc_split"      staged de Hoog planner: the per-step representation launch as one workgroup per 16-sample tile
 *                        (1, default; 60 vs 76 us per step at K = 16384, S = 33) or one wavefront per tile (0)
 *   "gru_coop"           stand-alone GRU encodes (nlc_gru_encode, nlc_model_forward, the two-launch planner bodies) with the
 *                        cooperative kernel -- one 16-window tile per workgroup, one gate chunk per wavefront, a third of
 *                        the latency -- 1 / 0; -1 = auto (default): up to 50 000 windows at hidden_units 128, 8 192 at 64, always at 256
 *   "gru_gemm"           EXPERIMENTAL.  0 (default): the GRU encoder's hidden-state GEMMs on FP64 MFMAs.  1: the same GEMMs as
 *                        int8-sliced fixed-point products on the INT8 matrix pipe (csrc/kernels_gru_i8.hip, nlc_i8gemm.h; models with
 *                        hidden_units = 128, the stand-alone encode launches that take the wave-sized form -- more than 50 000
 *                        windows, or all of them with gru_coop = 0; the one-launch planner body keeps its FP64 encoder role): GRU states and row-scaled weights as 54-bit fixed point in seven signed 8-bit digits, one
 *                        v_mfma_i32_16x16x64_i8 per digit pair, exact integer accumulation, FP64 recombination.  Latents
 *                        agree with the FP64 kernel's to 3e-16; against the exact product either form is within 5 x 2^-53 of
 *                        the row's sum of |w h| (tools/i8gemm_check.hip).  2.5 ms against 3.0 ms for BASELINE configs[1]'s encode
 *   "horizon_chunks"     Fourier planner, wave-per-tile body (more than 8192 local samples): the GRU encode in this many horizon
 *                        chunks on a low-priority stream of its own, each chunk's rollout launch behind its event, so that the
 *                        encoder of later steps runs beside the rollout of earlier ones (one wave of each kernel fits a SIMD:
 *                        216 + 288 VGPRs).  1 (default) = one encode launch, one rollout launch.  Same bits for every value;
 *                        measured 4.224 vs 4.270 ms at 3 chunks (K = 16384) -- the two kernels leave each other little to fill
 *                        (95 % / 85 % of the issue slots busy), and per-launch times of overlapped kernels say nothing about
 *                        either, so the default and the reported roofline stay with the two plain launches.
 *   "dehoog_chain"       de Hoog planner (single planner, hidden_units 128, 17 or 33 terms): 1 = the whole step chain as ONE
 *                        persistent launch (kernels_dehoog_chain.hip: a workgroup of eight waves owns 64 samples for all T
 *                        steps -- representation MLP of its four tiles in one pass, QD table one wavefront per dim, state / cost
 *                        tail; F_k stays with the CU that wrote it); 2 = the same with four waves per 32 samples, two workgroups
 *                        per CU; 0 = the staged 2 T + 1 launches below; -1 (default) = auto.  Same bits for every value.
 *                        AUTO (this knob and "dehoog_streams" both on auto, >= 8192 local samples): the planner MEASURES -- its
 *                        first commands run the candidate forms (staged on two streams, staged on one, persistent kernel) in
 *                        turns for at least four rounds and half a second, an event pair around the chain, and it keeps the
 *                        fastest.  On most boxes that is the staged path on two streams (155-159 planning steps/s at BASELINE
 *                        configs[4]'s size vs 151 on one stream and 150 for the persistent kernel); on a box whose streams do not
 *                        overlap two streams measure 134 and auto settles on one (profiles/r4_dehoog_chain.md).
 *                        ("dehoog_chain_phases": tools only, 1 / 2 = only the representation / only the QD phase of that kernel
 *                        runs -- a timing breakdown, meaningless results)
 *   "dehoog_streams"     staged planner (NLC_ILT_DEHOOG models; fixed Talbot / Stehfest models take the same path and the
 *                        same options): the population is cut into this many contiguous parts
 *                        whose per-step launches run on streams of their own -- one part's FP64-VALU-bound QD pass beside
 *                        another part's MFMA-bound representation launch; 0 = auto (2 from 8192 samples), 1 = one stream.
 *                        Same bits for every value.
 *   "dehoog_gru_chunks"  the same planner: GRU encode in this many horizon chunks on a further stream, beside the step chain
 *                        (cooperative kernel at reduced occupancy, "dehoog_gru_lds_pad" bytes of unused LDS); 0 / 1 (default):
 *                        one launch up front -- measured no faster (the chip is busy either way).
 *   "host_spin"          nlc_mppi_finish with action_host (single planner): 1 (default) = the merge kernel stores a sequence
 *                        number behind the action in pinned host memory and the host spins on that word (sub-microsecond
 *                        hand-over, one busy core for the length of a command); 2 = the host first sleeps through most of the
 *                        wait -- the shortest of its last eight waits predicts this one, it wakes "host_spin_margin_us"
 *                        (default 150; at least 15 % of the prediction) early -- and spins only for the rest, leaving the core
 *                        to the harness's other workers (run_exp_multi.py:145 fans out a Pool(12) on one GPU);
 *                        0 = hipStreamSynchronize (measured: the runtime busy-waits as well, and wakes 5-9 us later).
 *                        After the call ONLY the action is valid; the stream may still be finishing omega / cost_nz (they are
 *                        stream-ordered for every later call; nlc_synchronize before reading them from the host).
 *   "fused_blocks_per_cu"  fused body: the instance compiled for 3 (168 VGPRs) or 4 (128 VGPRs) workgroups per CU; 0 = auto
 *                        (3 while rollout chains sit on at most half of the CUs, else 4)
 *   "fused_inline"       fused body, single planner (E <= 1), bit mask: 1 = the importance-weight reduction (:210-216) runs
 *                        inside the launch (every rollout tile folds its 16 samples when their costs are final, the last one
 *                        folds the tiles); 2 = with device noise (rng == 1) and host-side state / action_buffer the sampling
 *                        and bounding (:319-328) run in the encoder role; 3 (default; 1 is accepted as 3) = both -- a command
 *                        is then two launches (this one + the merge after the shard all-gather) instead of five; 0 = separate
 *                        launches.  Same bits either way.
 *   "fused_spin_limit"   polls (~2 us each) before a waiting wave of the fused body gives up (default 2^18, ~0.5 s)
 *   "linear_fused"       NLC_ILT_FIXED_TALBOT / NLC_ILT_STEHFEST models: 1 (default) = the rollout runs on LIN instances of the
 *                        rollout kernels (two epilogue MFMAs per slot group, coefficient fragments folded for the constant
 *                        prediction time at nlc_mppi_configure); 0 = the staged path de Hoog models take
 *   "fused_keep_sync"    tools only: the merge kernel does not zero the fused body's sync block (tools/fused_debug.py reads
 *                        the launch's progress counters / timeline from it); the next command pays a memset instead
 *   "test_lin_coeff_scale"  tests only: the largest coefficient of the LIN rollout instances' folded w_re / t fragment is
 *                        scaled by this at the next nlc_mppi_configure (1 = off): the parity sweep's bound must catch 1 + 1e-6
 *   "fused_test_drop_tile"  tests only: the encoder tile with this ticket is never published (-1 = none): forces the
 *                        hand-off timeout and the re-run on the two-launch body
 *   "fused_chain_first_tiles"  encoder tiles every such workgroup encodes before its chain starts (-1 = auto: 1)
 *   "fused_partner_tiles"  the OTHER workgroups of that CU stop drawing encoder tiles after this many each and sleep until
 *                        the chain is done (an encoder wave beside a chain doubles the chain's step time); -1 = never,
 *                        -2 = auto (1 .. 4 with the share of CUs that walk a chain: fit to MI355X measurements)
 * The fused body assumes the device to itself: its rollout workgroups wait for encoder workgroups of the SAME launch, so
 * all of its workgroups must be resident at once (auto picks it only when at least two fit a CU).  Every wait is bounded: on
 * a time-out the launch marks its shard's partial row invalid (eta_r = -1), which travels through the shard all-gather, so
 * merge_kernel on EVERY rank skips the update and tells its host: every rank re-runs the command on the two-launch body inside
 * nlc_mppi_finish (library-side costs, host action pointer given; with a caller-owned collective the call returns NLC_AGAIN
 * for the second all-gather) -- no rank is left waiting in a collective.  Otherwise (device-resident action, cost callables)
 * the loss is reported as NLC_ERR_HIP by this or the next call, on every rank.  The ctx that timed out keeps to the
 * two-launch body from then on (setting "rollout_variant" again re-arms it).
 * Unknown names / out-of-range values: NLC_ERR_BAD_ARG. */
int nlc_set_option(nlc_ctx* ctx, const char* name, double value);
/* Read-only counters of the ctx's planner (ABI v9), so that a caller -- and the reference's deployment of several planner
 * processes on ONE GPU (run_exp_multi.py:145-165 fans out Pool(12) workers of K = 1000, config.py:52) -- can see which body
 * its commands run on and whether the fused body's bounded waits ever expired:
 *   "rollout_body"         body phase 1 of the LAST command ran on: 1 wave-per-tile | 2 latency-split | 3 fused one-launch
 *                          | 4 staged step chain (de Hoog / linear algorithms) | 5 persistent de Hoog chain | 6 oracle
 *                          | 7 Delta-t RNN | 8 NODE | 9 caller's callables; 0 = no command yet
 *   "fused_timeouts"       fused launches of this ctx that gave up waiting (each one: a re-run, or a lost command)
 *   "fused_fallbacks"      commands re-run on the two-launch body inside nlc_mppi_finish (give-up here OR on another rank)
 *   "fused_lost"           1 = this ctx has left the fused body for good ("rollout_variant" re-arms it)
 *   "last_giveup_command"  0-based index of the last command (count of nlc_mppi_rollout calls) that saw a give-up, -1 = none
 *   "commands"             nlc_mppi_rollout calls so far
 *   "comm_world", "comm_rank"  size / rank of the library-owned RCCL communicator (0 / -1 = none: nlc_comm_init not called)
 *   "fused_blocks_per_cu"  resident workgroups per CU the occupancy query returned for the fused kernel (-1 = not queried yet)
 *   "fused_spin_limit"     the option's current value
 *   "gru_gemm"             1 when encode launches run the int8-sliced kernel (option "gru_gemm" = 1 and a hidden_units = 128 model)
 *   "gru_i8_launches"      launches of the int8-sliced encoder kernel in this PROCESS so far (any ctx): the option alone does not say
 *                          that a launch took it -- cooperative and fused launches keep their FP64 encoder
 *   "model_nt3"            16-wide output tiles of the representation MLP's last layer as packed by nlc_set_model (the bench's
 *                          issued-flop count needs it), 0 = no model
 * Unknown names: NLC_ERR_BAD_ARG. */
int nlc_get_stat(nlc_ctx* ctx, const char* name, double* out);
/* device properties the bench reports next to its roofline numbers */
int nlc_device_info(nlc_ctx* ctx, char* name, int name_len, int* num_cus, int* clock_mhz, double* hbm_gib);

/* ---- ILT: torchlaplace.laplace_reconstruct (external; call sites w_nl.py:137-144,
 *      w_latent_ode.py:88-94).  Parity unpinned vs upstream torchlaplace: alpha/tol/scale and the
 *      [theta_s | phi_s | p] input order are the recalled defaults, exposed here as parameters. ---- */
typedef struct {
  int32_t algo;  /* NLC_ILT_* */
  int32_t terms; /* S = ilt_reconstruction_terms (de Hoog: odd, 2M+1 <= 33; Stehfest: even <= 20) */
  double alpha;  /* fourier 1e-3, dehoog 1e-10 (unused by the linear algorithms, as are tol and scale) */
  double tol;    /* 10*alpha */
  double scale;  /* 2.0 */
} nlc_ilt_desc;

/* Contour evaluation + Riemann-sphere projection + concat: rows [theta_s(0..S-1) | phi_s(0..S-1) | p]
 * for every (batch row b, time point j).  t_dev is (B*Tt) if t_batched else (Tt) shared by all rows.
 * out_dev: (B, Tt, 2S+P). */
int nlc_ilt_rep_inputs(nlc_ctx* ctx, const nlc_ilt_desc* desc, const double* p_dev, const double* t_dev,
                       int t_batched, int64_t B, int64_t Tt, int P, double* out_dev);
/* Sphere -> complex map and line integral: theta_dev, phi_dev (N, d, S) rep-func outputs, t_dev (N)
 * -> x_dev (N, d).  HBM-bound streaming kernel for Fourier; algorithmic bytes (2dS+d)*8 per point. */
int nlc_ilt_reconstruct(nlc_ctx* ctx, const nlc_ilt_desc* desc, const double* theta_dev, const double* phi_dev,
                        const double* t_dev, int64_t N, int d, double* x_dev);

/* Backward of nlc_ilt_reconstruct with respect to the representation-function outputs, every algorithm: the
 * reference trains the rep func THROUGH torchlaplace.laplace_reconstruct by autograd (train_utils.py:388-407 ->
 * w_nl.py:137-144).  grad_x_dev (N, d) upstream gradient -> grad_theta_dev, grad_phi_dev (N, d, S).  No gradient
 * with respect to t.  Fourier / fixed Talbot / Stehfest: HBM-bound streaming kernels, algorithmic bytes 4dS*8 per point.
 * de Hoog: reverse mode through the quotient-difference table, taped in stream-ordered scratch of the call
 * (hipMallocAsync / hipFreeAsync on the bound stream: 7.5 KB per row in flight at S = 33 -- 466 complex values: the q columns,
 * the even e columns, the continued fraction's recurrence -- for at most 1024 resident wavefronts of 64 rows: 0.49 GB). */
int nlc_ilt_reconstruct_backward(nlc_ctx* ctx, const nlc_ilt_desc* desc, const double* theta_dev,
                                 const double* phi_dev, const double* t_dev, const double* grad_x_dev, int64_t N,
                                 int d, double* grad_theta_dev, double* grad_phi_dev);

/* ---- model: NeuralLaplaceModel (w_nl.py:66-145), ReverseGRUEncoder (:14-29),
 *      LaplaceRepresentationFunc (:32-63) ------------------------------------------------------- */
typedef struct {
  int32_t d;   /* state_dim (obs dim) */
  int32_t nin; /* GRU input dim = action_dim + encode_obs_time (w_nl.py:19-20) */
  int32_t h;   /* hidden_units (GRU hidden = h/2, w_nl.py:92) */
  nlc_ilt_desc ilt;
  double time_div; /* ts_pred is divided by this: dt*8 if normalize&&normalize_time else 1 (w_nl.py:122) */
  double state_mean[NLC_MAX_D], state_std[NLC_MAX_D];       /* identity (0,1) if !normalize */
  double action_mean[NLC_MAX_NIN], action_std[NLC_MAX_NIN]; /* (0,3) if !normalize (w_nl.py:129) */
} nlc_model_desc;

/* weights_host: float64 blob, tensors in the reference's state_dict order:
 *   gru.weight_ih_l0 (3g,nin) weight_hh_l0 (3g,g) bias_ih_l0 (3g) bias_hh_l0 (3g)
 *   gru.weight_ih_l1 (3g,g)   weight_hh_l1 (3g,g) bias_ih_l1 (3g) bias_hh_l1 (3g)
 *   linear_out.weight (2,g) linear_out.bias (2)
 *   linear_tanh_stack.0.weight (h,2S+d+2) .0.bias (h) .2.weight (h,h) .2.bias (h)
 *   .4.weight (2dS,h) .4.bias (2dS)
 * n_doubles must equal nlc_model_blob_size(desc).  The ctx repacks them into MFMA fragment order.
 * desc->ilt.algo: every implemented algorithm runs nlc_gru_encode, nlc_rep_func, nlc_model_forward and the planner
 * (NLC_DYN_NL; Fourier: fused rollout kernel, the others: staged all-HIP path); nlc_model_forward_const_t is Fourier only. */
int64_t nlc_model_blob_size(const nlc_model_desc* desc);
int nlc_set_model(nlc_ctx* ctx, const nlc_model_desc* desc, const double* weights_host, int64_t n_doubles);
/* ReverseGRUEncoder.forward on the model's normalised input: window_dev (N, B, nin) raw
 * (un-normalised) actions, B = window length (action_buffer_size, config.py:58) -> out_dev (N, 2) */
int nlc_gru_encode(nlc_ctx* ctx, const double* window_dev, int64_t N, int B, double* out_dev);
/* NeuralLaplaceModel.forward: obs_dev (N,d), window_dev (N,B,nin), ts_dev (N) raw ts_pred
 * -> out_dev (N,d) predicted state difference.  ws_dev: scratch of nlc_model_workspace_bytes(N). */
int64_t nlc_model_workspace_bytes(nlc_ctx* ctx, int64_t N);
int nlc_model_forward(nlc_ctx* ctx, const double* obs_dev, const double* window_dev, const double* ts_dev,
                      int64_t N, int B, double* out_dev, void* ws_dev);
/* The same with ONE query time for every row (what the harness closure passes: ts_pred = dt, mppi_with_model.py:74,
 * 120): the 2S sphere-coordinate inputs of the representation MLP are then constants and are folded into its first
 * bias on the host (once per value of ts_pred), as nlc_mppi_configure does for the fused planner.  Fourier models. */
int nlc_model_forward_const_t(nlc_ctx* ctx, const double* obs_dev, const double* window_dev, double ts_pred, int64_t N,
                              int B, double* out_dev, void* ws_dev);
/* a8 alone: LaplaceRepresentationFunc.forward (w_nl.py:55-63) on N explicit input rows
 * rep_in_dev (N, 2S + d + 2) = [theta_s (S) | phi_s (S) | latent p (d+2)] -- whatever sphere coordinates the caller
 * supplies, as torchlaplace hands them to the module -- -> theta_dev, phi_dev (N, d, S) with theta = pi tanh(.),
 * phi = (pi/2) tanh(.) (:59-62).  Same MFMA kernel as the de Hoog path's representation stage. */
int nlc_rep_func(nlc_ctx* ctx, const double* rep_in_dev, int64_t N, double* theta_dev, double* phi_dev);

/* ---- baseline models: DeltaTRNN (train_utils.py:589-631; factory :56-74) and RNN (:550-586; factory :77-98);
 *      rnn_hidden_units config.py:43 ------
 * out, _ = GRU(nin -> hidden, 1 layer, batch_first)(window_n); dx = linear_out(cat(out[:, -1], obs_n, ts_n)). */
typedef struct {
  int32_t d;      /* state_dim */
  int32_t nin;    /* GRU input dim = action_dim (+1 for an encode_obs_time model; forward API only) */
  int32_t hidden; /* hidden_units: 64, 128 or 160 */
  int32_t time_input; /* 1: DeltaTRNN (linear_out sees ts); 0: the plain RNN baseline (train_utils.py:550-586), whose
                       * linear_out is (d, H+d) and which ignores ts_pred */
  /* The caller resolves the reference's branch structure (train_utils.py:618-626; the `else` of the raw-input branch
   * belongs to `if self.normalize_time`): normalised branch -> the model's buffers and time_div = dt*8; raw branch
   * (normalize_time False) -> state (0, 1), action (0, 3), time_div 1. */
  double time_div;
  double state_mean[NLC_MAX_D], state_std[NLC_MAX_D];
  double action_mean[NLC_MAX_NIN], action_std[NLC_MAX_NIN];
} nlc_rnn_desc;
/* weights_host: float64 blob in the reference's state_dict order:
 *   gru.weight_ih_l0 (3H,nin) gru.weight_hh_l0 (3H,H) gru.bias_ih_l0 (3H) gru.bias_hh_l0 (3H)
 *   linear_out.weight (d, H+d+time_input) linear_out.bias (d) */
int64_t nlc_rnn_blob_size(const nlc_rnn_desc* desc);
int nlc_set_rnn_model(nlc_ctx* ctx, const nlc_rnn_desc* desc, const double* weights_host, int64_t n_doubles);
/* DeltaTRNN.forward: obs_dev (N,d), window_dev (N,B,nin) raw actions, ts_dev (N) raw ts_pred -> out_dev (N,d)
 * predicted state difference.  ws_dev: N*d doubles of scratch. */
int nlc_rnn_forward(nlc_ctx* ctx, const double* obs_dev, const double* window_dev, const double* ts_dev, int64_t N,
                    int B, double* out_dev, void* ws_dev);

/* ---- baseline model: NODE (train_utils.py:664-724; ODE function xOdeFuncInXAndU :637-661; factory :101-125;
 * node_hidden_units 270, node_augment_dim 1, node_method "euler": config.py:40-42).
 *   x = cat((obs - mean)/std, zeros(augment_dim)); u = window[:, -1, :] (raw);
 *   out = odeint(f(., u), x, [0, ts_pred[0] / time_div], method="euler", step_size)[-1][:, :d]
 * torchdiffeq's fixed-grid Euler solver is restated (grid t_k = k*step_size, last point = the end time):
 * PARITY UNPINNED vs upstream torchdiffeq (absent offline); everything else is pinned against the reference classes. */
typedef struct {
  int32_t d;           /* state_dim */
  int32_t nu;          /* action_dim */
  int32_t hidden;      /* hidden_units, <= 272 */
  int32_t augment_dim; /* d + augment_dim <= 8, d + augment_dim + nu <= 12 */
  double time_div;     /* dt*8 if normalize_time else 1 (train_utils.py:701-702) */
  double step_size;    /* options["step_size"] = 0.05 (:722) */
  double state_mean[NLC_MAX_D], state_std[NLC_MAX_D]; /* (0, 1) if !normalize */
} nlc_node_desc;
/* weights_host: float64 blob in state_dict order: x_ode_func_in_x_and_u.linear_tanh_stack.{0,2,4}.{weight,bias}:
 *   (H, d+aug+nu) (H) (H, H) (H) (d+aug, H) (d+aug) */
int64_t nlc_node_blob_size(const nlc_node_desc* desc);
int nlc_set_node_model(nlc_ctx* ctx, const nlc_node_desc* desc, const double* weights_host, int64_t n_doubles);
/* NODE.forward: obs_dev (N,d), action_dev (N,nu) = window[:, -1, :], ts_pred = the FIRST row's raw prediction time
 * (the reference integrates every row to ts_pred[0], :719) -> out_dev (N,d) integrated normalised state. */
int nlc_node_forward(nlc_ctx* ctx, const double* obs_dev, const double* action_dev, double ts_pred, int64_t N,
                     double* out_dev);

/* ---- planner: MPPIDelay (planners/mppi_delay.py:54-381) ------------------------------------- */
typedef struct {
  int64_t K;        /* samples owned by THIS ctx (its shard of the population) */
  int64_t K_global; /* whole population (== K on one GPU) */
  int64_t k_offset; /* global index of this shard's first sample (Philox counters, sample_null_action) */
  int32_t T;        /* horizon */
  int32_t nu;       /* action dims */
  int32_t d;        /* nx */
  int32_t B;        /* rows of action_buffer */
  double lambda_;
  double u_scale;
  int32_t has_bounds; /* u_min/u_max given (mppi_delay.py:143-153) */
  double u_min[NLC_MAX_NU], u_max[NLC_MAX_NU];
  double u_init[NLC_MAX_NU];
  double noise_mu[NLC_MAX_NU];
  double noise_sigma[NLC_MAX_NU * NLC_MAX_NU];     /* covariance, row-major nu x nu */
  double noise_sigma_inv[NLC_MAX_NU * NLC_MAX_NU]; /* torch.inverse(noise_sigma) (:157) */
  double noise_chol[NLC_MAX_NU * NLC_MAX_NU];      /* lower Cholesky factor (device RNG only) */
  int32_t sample_null_action; /* :322-323 */
  int32_t noise_abs_cost;     /* :329-330 */
  int32_t u_per_command;      /* :217-220 */
  int32_t dynamics;           /* NLC_DYN_* */
  int32_t env;                /* NLC_ENV_* running cost (and oracle dynamics) */
  int32_t delay;              /* oracle dynamics: action applied = window[-(delay+1)] */
  int32_t friction;           /* oracle cartpole only */
  int32_t E;                  /* episodes: independent planning problems batched in this ctx, each with its own
                               * state, action_buffer, U and K samples (0 or 1 = one problem = the reference's
                               * MPPIDelay; > 1 = the dataset collector's many episodes,
                               * mppi_dataset_collector.py:224-321,402-424, planned side by side) */
  int32_t cost_external;      /* 1: the dynamics run fused but the caller evaluates running_cost / terminal cost itself
                               * (arbitrary harness closures, e.g. the state_constraint / change_goal branches of
                               * mppi_with_model.py:145-171): nlc_mppi_rollout leaves only the perturbation cost
                               * (:343-344) in buf->cost_total and stops before the weights; the caller adds its cost
                               * (buf->states must be given) and calls nlc_mppi_weights.  env is then used by oracle
                               * dynamics only and may be -1 with NL dynamics. */
  double ts_pred;             /* raw dt handed to the dynamics (mppi_with_model.py:74) */
} nlc_mppi_desc;

/* caller-owned device buffers the kernels read/write (the Python mirror keeps them as the public
 * attributes .noise .perturbed_action .states .actions .cost_total .cost_total_non_zero .omega).
 * With E > 1 episodes every buffer gains a leading E: per-sample ones are (E,K,...), partials
 * (E, 2+T*nu), action (E, u_per_command*nu); U is (E,T,nu). */
typedef struct {
  double* noise;      /* (K,T,nu) in: raw N(mu,Sigma) draw when rng == 0; out: bounded noise (:328) */
  double* perturbed;  /* (K,T,nu) out: bounded perturbed action, normalised units (:325-326) */
  double* states;     /* (K,T,d)  out, may be NULL */
  double* actions;    /* (K,T,nu) out = perturbed (actions/u_scale, :340), may be NULL */
  double* cost_total; /* (K) out */
  double* cost_nz;    /* (K) out: exp(-(cost-beta)/lambda) */
  double* omega;      /* (K) out: cost_nz/eta, may be NULL */
  double* partials;   /* (2+T*nu) out: this shard's (beta_r, eta_r, S_r[t,j]) */
  double* action;     /* (u_per_command*nu) out: the returned action, on the device; may be NULL */
  void* workspace;    /* nlc_mppi_workspace_bytes() bytes of scratch */
} nlc_mppi_buffers;

/* NLC_DYN_NL: the model's GRU input dim must be nu, or nu+1 for an encode_obs_time model -- the rollout then
 * appends the constant time channel B-1 .. 0 the harness closure builds (mppi_with_model.py:110-119).
 * NLC_DYN_DTRNN: needs nlc_set_rnn_model first; GRU input dim == nu (the closure adds the time channel for
 * model_name == "nl" only).  NLC_DYN_NODE: needs nlc_set_node_model first. */
int nlc_mppi_configure(nlc_ctx* ctx, const nlc_mppi_desc* desc);
int64_t nlc_mppi_workspace_bytes(nlc_ctx* ctx);
int nlc_mppi_set_U(nlc_ctx* ctx, const double* U_host); /* (E,T,nu) control sequence(s), :161-164 */
int nlc_mppi_get_U(nlc_ctx* ctx, double* U_host);
/* Phase 1 of command(): shift U (:199-200), sample/perturb/bound (:319-335), hoisted GRU encode,
 * T-step rollout + running cost (:232-313), perturbation cost (:343-344), and this shard's
 * softmax partials (beta_r, eta_r, S_r) into buf->partials.
 *   state_host: (d) or (K,d) if state_per_sample (:243-246); action_buffer_host: (B,nu).
 *   E > 1: state (E,d) or (E,K,d), action_buffer (E,B,nu); both may then be host OR device pointers.
 *   rng: 0 = buf->noise holds the caller's raw draw; 1 = device Philox4x32-10(seed, counter). */
int nlc_mppi_rollout(nlc_ctx* ctx, const double* state_host, int state_per_sample,
                     const double* action_buffer_host, const nlc_mppi_buffers* buf, int rng, uint64_t seed,
                     uint64_t counter);
/* NLC_DYN_EXTERNAL / cost_external: nlc_mppi_rollout stops after the perturbation / after the rollout; once the
 * caller has completed buf->cost_total (rollout cost + perturbation cost, :339-344) this computes the softmax
 * partials. */
int nlc_mppi_weights(nlc_ctx* ctx, const nlc_mppi_buffers* buf);
/* Phase 2: merge G shard partials (gathered_dev: (G, E, 2+T*nu); pass buf->partials and G=1 on one GPU; NULL: gather
 * them with the ctx's own communicator, see the multi-GPU note below),
 * omega, U[t] += sum_k omega_k noise[k,t] (:210-216) and return action = U[:u_per_command]*u_scale
 * (:217-224) into action_host (E*u_per_command*nu) -- the call waits until the ACTION is in host memory (see "host_spin":
 * cost_nz / omega may still be in flight on the stream; nlc_synchronize before reading them from the host) -- and/or into
 * buf->action on the device.  With action_host == NULL nothing is copied back and the call does not synchronise.
 * Returns NLC_AGAIN (> 0, not an error) when a sharded command had to be re-run and the caller owns the collective. */
int nlc_mppi_finish(nlc_ctx* ctx, const double* gathered_dev, int G, int rank, const nlc_mppi_buffers* buf,
                    double* action_host);
/* Multi-GPU: the one exchange of a K-sharded command is the all-gather of buf->partials ((2+T*nu) doubles per rank and
 * episode) between nlc_mppi_rollout and nlc_mppi_finish (reference reduction :210-216).  Two ways:
 *   (1) the caller brings the collective and passes `gathered_dev`: torch.distributed over RCCL in the Python mirror
 *       (sharding.py, the default there), MPI_Allgather / its own RCCL communicator for a C caller -- the library does not
 *       compete with the host framework for the collective stream, ranks or device selection;
 *   (2) the library's own communicator (SURVEY 8b's sketch): rank 0 draws an id with nlc_comm_unique_id and ships the
 *       NLC_COMM_ID_BYTES bytes to every rank by any means, every rank calls nlc_comm_init on its ctx (its device), and
 *       nlc_mppi_finish with gathered_dev == NULL then runs ONE ncclAllGather on the command's own stream before the
 *       merge -- no host hop between the rollout and the action.  RCCL is bound at run time (dlopen of librccl.so.1; the
 *       copy already in the process is used if there is one); without it these return NLC_ERR_UNSUPPORTED.
 * nlc_comm_unique_id has no ctx: its error text is returned by nlc_last_error(NULL). */
#define NLC_COMM_ID_BYTES 128
int nlc_comm_unique_id(void* id_out);
int nlc_comm_init(nlc_ctx* ctx, int rank, int world, const void* id);
int nlc_comm_destroy(nlc_ctx* ctx);
/* Collective: every rank all-gathers its rank number over the communicator and checks the result on the host (the Python
 * mirror runs it once after nlc_comm_init and falls back to torch.distributed's collective if any rank fails). */
int nlc_comm_self_test(nlc_ctx* ctx);

/* ---- env side of the evaluation loop: the reference steps ONE env per process on the host,
 * step_env (mppi_with_model.py:193-216) = get_action (delay buffer, :25-28) + env.integrate_system(2, g, s0)
 * (base_env.py:136-173; solver "euler" overlay.py:39, ts = [0, dt]: one Euler step of torch_rhs on the reduced
 * state) + get_obs (:83-89).  Here E independent envs advance on the device, next to a batched planner (E > 1), so a
 * control step needs no host round trip.
 *   state_dev (E, n) reduced states (cartpole [x, xdot, theta, thetadot], pendulum [theta, thetadot], acrobot
 *   [theta1, theta2, v1, v2]), updated in place; action_buffer_dev (E, B, nu) rolled in place, the new action
 *   appended, row -(delay+1) applied; action_dev (E, nu); obs_dev (E, d) trig observation of the new state;
 *   reward_dev (E) diff_reward(new state, applied action), may be NULL. */
int nlc_env_step(nlc_ctx* ctx, int env, int friction, double dt, int delay, int64_t E, int B, int nu,
                 double* state_dev, double* action_buffer_dev, const double* action_dev, double* obs_dev,
                 double* reward_dev);
/* get_obs (base_env.py:83-89): obs_dev (E, d) = torch_transform_states(state_dev (E, n)) */
int nlc_env_obs(nlc_ctx* ctx, int env, int64_t E, const double* state_dev, double* obs_dev);

/* ---- in-library kernel timing (hipEvent pairs on the launch stream) ------------------------- */
int nlc_profile_enable(nlc_ctx* ctx, int on);
int nlc_profile_reset(nlc_ctx* ctx);
int nlc_profile_count(nlc_ctx* ctx);
int nlc_profile_read(nlc_ctx* ctx, int idx, char* name, int name_len, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* NLC_H_ */
