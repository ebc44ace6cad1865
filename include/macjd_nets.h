/*
 * macjd_nets.h — C-ABI of the fused agent-side kernels in libmacjd_hip.so (MI355X / gfx950).
 *
 * The reference has no native layer; the interfaces replaced here are Python call sites:
 *   macjd_qhead_select   BasicMAC.select_actions' multi-pass loop + mask + epsilon-greedy + gather
 *                        (reference core/mac.py:109-164, utils/action_selectors.py:15-62) and the
 *                        per-action loop of QMixLearner._get_all_action_q_values_and_params
 *                        (core/qmix.py:256-274), both built on RNNAgent.get_q_value_for_action
 *                        (core/networks.py:131-180);
 *   macjd_gru_sequence   the learner's `for t in range(max_seq_len)` GRU unroll
 *                        (core/qmix.py:241-253 -> core/networks.py:88-114);
 *   macjd_mixer_tail_*   QMixer.forward after the hyper-network GEMMs, and its backward
 *                        (core/networks.py:283-315);
 *   macjd_td_loss        TD target + masked MSE + its gradient + logged means (core/qmix.py:155,190-194,212-213).
 * Device pointers, element strides, no torch / HIP types; asynchronous on the given stream.
 */
#ifndef MACJD_NETS_H
#define MACJD_NETS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Q(h, a, P[:,a]) for every discrete action a in one pass, optionally followed by availability
 * masking, epsilon-greedy choice and the gather of the chosen power.
 *
 * The Q-head's first layer is W1 [H, H+A+1] applied to cat([h, onehot(a), P_a]) (networks.py:75-79,
 * 171-174).  The caller supplies base = W1[:, :H] h + b1 (one GEMM, [N,H]); the kernel adds the
 * action column W1[:, H+a] and the rank-1 power term W1[:, H+A] * P_a, applies ReLU and the second
 * layer (w2, b2), for all a, without materialising [N, A, H].
 * Row n of base / P_all / Q is (env e, agent j) with n = e * n_agents + j.
 */
typedef struct macjd_qhead_io {
    int64_t n_rows;          /* N = E * n_agents */
    int32_t H, A, n_agents, greedy_only; /* greedy_only != 0: test_mode (argmax, no exploration) */
    const float* base;  int64_t base_ld;   /* [N,H] row stride in elements */
    const float* P_all; int64_t p_ld;      /* [N,A] actor output (networks.py:116-129) */
    const float* W1;    int64_t w1_ld;     /* fc2_q_head.0.weight [H, H+A+1], row stride */
    const float* w2;                       /* fc2_q_head.2.weight [H] */
    const float* b2;                       /* fc2_q_head.2.bias  [1] */
    float* Q;           int64_t q_ld;      /* optional out [N,A]: UNMASKED Q-values */
    /* ---- optional selection stage (T_out32 or T_out64 non-NULL) ---- */
    const void* avail;  int32_t avail_elem_size, reserved; /* optional mask, int32 (4) or int64 (8) elements */
    int64_t av_se, av_sj, av_sa;           /* avail element strides over (env, agent, action) */
    float epsilon;      float reserved2;   /* exploration probability (action_selectors.py:30-32) */
    uint64_t seed, counter;                /* Philox key / call counter for the exploration draws */
    /* optional device-side sources, for launches replayed from a captured HIP graph (kernel arguments
       are frozen at capture): *eps_dev replaces epsilon, *counter_dev is ADDED to counter */
    const float* eps_dev;
    const uint64_t* counter_dev;
    int32_t* T_out32;                      /* optional chosen action, int32 */
    int64_t* T_out64;                      /* optional chosen action, int64 */
    int64_t t32_se, t32_sj, t64_se, t64_sj;/* element strides over (env, agent) */
    float* P_out;       int64_t po_se, po_sj; /* optional chosen power = P_all[n, T] (mac.py:151-164) */
    /* ---- Double-DQN helpers of the learner (reference core/qmix.py:138-147), contiguous [N] ----
       argmax_out[n] = first arg-max of the UNMASKED Q-values (the reference applies no mask there);
       q_gather_out[n] = Q[n, gather_idx[n]] (the target network's value of the eval network's choice). */
    int64_t* argmax_out;
    const int64_t* gather_idx;
    float* q_gather_out;
} macjd_qhead_io;

int macjd_qhead_select(const macjd_qhead_io* io, void* hip_stream);

/*
 * All T steps of the GRU recurrence for B*J independent sequences, for up to two networks (eval and
 * target) in one launch.  Replaces the learner's `for t in range(max_seq_len): h = agent.forward(...)`
 * unroll (reference core/qmix.py:241-253 -> core/networks.py:88-114 -> torch.nn.GRUCell), inference
 * only (the reference never back-propagates through it, see SURVEY.md section 8a note 4).
 *
 * gi = W_ih x_t + b_ih is time-parallel and supplied by the caller for every step ([B,T,J,3H],
 * contiguous, gate order r,z,n); the kernel evaluates  gh = W_hh h + b_hh,  r = s(gi_r + gh_r),
 * z = s(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),  h' = (h - n) z + n  and stores every h' ([B,T,J,H]).
 * H must be 64 or 128.
 */
typedef struct macjd_gru_io {
    int32_t n_nets, B, T, J, H, reserved;   /* n_nets = 1 or 2; reserved != 0 ("gi_static"): gi is [B,1,J,3H], the
                                               same input transform at every step (static observation) */
    const float* gi[2];    /* [B,T,J,3H] */
    const float* w_hh[2];  /* rnn.weight_hh [3H,H] row-major */
    const float* b_hh[2];  /* rnn.bias_hh [3H] */
    const float* h0[2];    /* optional initial state [B,J,H]; NULL = zeros (mac.init_hidden) */
    float* h_out[2];       /* [B,T,J,H] */
    int64_t h0_sb[2];      /* element stride between batch entries of h0 (0 = J*H, contiguous): lets the caller pass
                              step 0 of a stored [B,T+1,J,H] hidden-state tensor without copying it */
    /* Static observation, input transform computed IN the kernel (gi[] may then be NULL): sequence (b, j) reads its
       observation row obs + row(b) * obs_sb + j * obs_sj (row(b) = obs_index[b] if obs_index else b — e.g. the sampled
       episodes' step-0 rows straight out of the replay ring, no gather in front of the scan) and evaluates
       gi = W_ih ReLU(fc1 obs + b_fc1) + b_ih once (reference core/networks.py:96-100), used at every step. */
    const float* obs;      int64_t obs_sb, obs_sj;   int32_t S, reserved2;
    const int64_t* obs_index;                        /* optional [B] */
    const float* fc1_w[2]; const float* fc1_b[2];    /* fc1.weight [H,S], fc1.bias [H] */
    const float* w_ih[2];  const float* b_ih[2];     /* rnn.weight_ih [3H,H], rnn.bias_ih [3H] */
    /* optional, with obs: the frozen actor chain of the same observation row (core/networks.py:116-129),
       p_out[net][(b * J + j) * A + a] = sigmoid(L3 ReLU(L2 ReLU(L1 x))) — the continuous parameter of every action,
       one row per SEQUENCE (it is the same at every step).  Ah <= 256, A <= 64. */
    float* p_out[2];
    const float* act_w[2][3]; const float* act_b[2][3];   /* actor.{0,2,4}.{weight,bias}: [Ah,S], [Ah,Ah], [A,Ah] */
    int32_t Ah, A;
} macjd_gru_io;

int macjd_gru_sequence(const macjd_gru_io* io, void* hip_stream);

/*
 * QMixer.forward after the hyper-network GEMMs (reference core/networks.py:283-315), per row m:
 *   w1 = clamp(w1_raw, 0, 5) [J,Em]   b1 = clamp(b1_raw, -5, 5) [Em]   wf = clamp(wf_raw, 0, 5) [Em]
 *   v = clamp(v_raw, -5, 5)           hid = q . w1 + b1                y = ELU(hid) . wf + v
 * (torch.bmm([M,1,J],[M,J,Em]) / F.elu / torch.bmm([M,1,Em],[M,Em,1]) in the reference) and its backward
 * (clamp passes the gradient where min <= x <= max, like torch.clamp).  All tensors contiguous float32 (b1_raw may
 * have a row stride).
 * Forward needs q, w1_raw, b1_raw, wf_raw, v_raw, y.  Backward additionally gy and the five gradient outputs.
 */
typedef struct macjd_mixer_io {
    int64_t M;              /* rows = B * (T-1) */
    int32_t J, Em;          /* agents, mixing_embed_dim */
    const float* q;         /* [M,J]    */
    const float* w1_raw;    /* [M,J*Em] hyper_w_1 output   */
    const float* b1_raw;    /* [M,Em]   hyper_b_1 output   */
    const float* wf_raw;    /* [M,Em]   hyper_w_final output */
    const float* v_raw;     /* [M]      V output           */
    float* y;               /* [M]      Q_tot (forward out) */
    const float* gy;        /* [M]      dL/dy (backward in) */
    float* gq;              /* [M,J]    (backward outs)    */
    float* gw1_raw;         /* [M,J*Em] */
    float* gb1_raw;         /* [M,Em]   */
    float* gwf_raw;         /* [M,Em]   */
    float* gv_raw;          /* [M]      */
    int64_t b1_ld;          /* row stride of b1_raw in elements (0 = Em): b1_raw may be a column block of the merged
                               first-layer output */
} macjd_mixer_io;

int macjd_mixer_tail_forward(const macjd_mixer_io* io, void* hip_stream);
int macjd_mixer_tail_backward(const macjd_mixer_io* io, void* hip_stream);

/*
 * TD target + masked mean-squared TD error of QMixLearner.train (reference core/qmix.py:155,190-194):
 *   target = r + gamma (1 - terminated) tq ;  loss = sum((filled (y - target))^2) / sum(filled)
 * over M = B * Tm1 (batch, step) pairs, plus dL/dy = 2 filled (y - target) / sum(filled) and the two logged means
 * (eval_qtot_avg = mean(y), target_qtot_avg = mean(target), qmix.py:212-213).  One single-workgroup launch.
 * y / tq / reward / terminated / filled are addressed with batch (and step) element strides so that slices of
 * longer [B,T(+1),1] tensors need no copy; gy may be a full-length row with the unused tail zeroed.  terminated / filled are 1-byte bools.
 */
typedef struct macjd_tdloss_io {
    int32_t B, Tm1;
    float gamma, reserved;
    const float* y;            /* eval Q_tot:   element (b, t) at y[b * y_sb + t],  t < Tm1 */
    const float* tq;           /* target Q_tot: element (b, t) at tq[b * tq_sb + t]         */
    int64_t y_sb, tq_sb;       /* batch strides (Tm1 for contiguous [B*Tm1] tensors)        */
    int64_t gy_sb, gy_cols;    /* gy row pitch and row length: columns Tm1..gy_cols-1 are written as zeros */
    const float* reward;       int64_t r_sb, r_st;
    const uint8_t* terminated; int64_t t_sb, t_st;
    const uint8_t* filled;     int64_t f_sb, f_st;
    float* stats;              /* [4] out: loss, mean(y), mean(target), sum(filled) */
    float* gy;                 /* out: dL/dy, element (b, t) at gy[b * gy_sb + t]; NULL: the logged sums only — stats[0..2]
                                  are written, stats[3] is left alone, no gradient (macjd_mixer_fused_backward_td forms it) */
} macjd_tdloss_io;

int macjd_td_loss(const macjd_tdloss_io* io, void* hip_stream);
/* out[0] = sum of the loss mask (io->filled over B x Tm1; the other fields are not read): the only global quantity the
   gradient of the loss needs — see macjd_mixer_fused_backward_td. */
int macjd_td_mask_sum(const macjd_tdloss_io* io, float* out, void* hip_stream);

/*
 * Gradient clipping + Adam over ONE flat parameter vector (reference core/qmix.py:199-200:
 * torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step()):
 *   total_norm = ||grad||_2 ; coef = min(1, max_norm / (total_norm + 1e-6)) ; g = coef * grad
 *   step += 1 ; m = m + (1 - b1)(g - m) ; v = b2 v + (1 - b2) g g
 *   param -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * param / grad / exp_avg / exp_avg_sq are contiguous float32 [n]; step is a float32 device scalar (torch's
 * capturable Adam keeps it on the device), read-modify-written by the kernel; grad_norm (out) = total_norm.
 * `partials` is caller-provided scratch of >= 256 floats.  Two launches.
 */
typedef struct macjd_adam_io {
    int64_t n;
    float lr, beta1, beta2, eps, max_norm, reserved;
    float* param; float* grad /* scaled in place by coef, like clip_grad_norm_ */; float* exp_avg; float* exp_avg_sq;
    float* step; float* grad_norm; float* partials;
} macjd_adam_io;

int macjd_clip_adam_step(const macjd_adam_io* io, void* hip_stream);

/*
 * Draw of the NEXT update's episodes on the device: idx_out[0..n) = n distinct indices, uniform over [0, *n_stored) —
 * what the reference's buffer does on the host with np.random.choice(current_size, batch, replace=False)
 * (utils/replay_buffer.py:89) — as the first n images of a keyed pseudo-random permutation of [0, *n_stored):
 * an 8-round Feistel network on the ceil(log2 N) bits of an index, cycle-walked back into [0, N), round keys =
 * Philox4x32-10(counter = *counter, key = seed).  *counter advances by one per draw, so a sequence of draws is a
 * function of (seed, first counter value) only.  A host-drawn batch needs an index upload between two updates, which
 * sits on the serial chain of the replayed graphs (~10 us of a ~190 us step); this draw rides at the end of the previous
 * update's last launch (macjd_clip_adam_step_sample), or stands alone (macjd_sample_episodes: first draw, redraw after
 * the population changed).  *n_stored < n: idx_out[i] = i mod *n_stored (callers fall back to the host path before).
 */
typedef struct macjd_sampler_io {
    int64_t* idx_out;          /* [n] */
    int32_t n, reserved;
    const int32_t* n_stored;   /* device scalar: stored episodes (the population) */
    int64_t* counter;          /* device scalar: draws made so far */
    uint64_t seed;
} macjd_sampler_io;

int macjd_sample_episodes(const macjd_sampler_io* io, void* hip_stream);
/* macjd_clip_adam_step followed, inside its last launch, by the draw of the next update's episodes (next may be NULL) */
int macjd_clip_adam_step_sample(const macjd_adam_io* io, const macjd_sampler_io* next, void* hip_stream);

/*
 * Weight and bias gradient of y = x W^T + b over K rows:  dW[M,N] = gout[K,M]^T x inp[K,N],  db[M] = sum_k gout[k,:]
 * (the backward of torch.nn.Linear for its parameters, core/qmix.py:198 loss.backward()).  On this path the outputs
 * are tiny (e.g. [192,128]) and K is 3e3..1e4 rows, so the reduction is split over K: every workgroup computes one
 * 64x64 output tile for one 128-row K-chunk on exact-f32 MFMA into `workspace`, a second launch sums the chunks in
 * fixed order (deterministic).  workspace >= macjd_linear_wgrad_workspace_floats(K, M, N) floats.
 */
typedef struct macjd_wgrad_io {
    int64_t K;
    int32_t M, N;
    const float* gout; int64_t gout_ld;   /* [K,M] row stride (elements) */
    const float* inp;  int64_t inp_ld;    /* [K,N] */
    float* dW;         int64_t dw_ld;     /* [M,N] out */
    float* db;                            /* [M] out, optional */
    float* workspace;
    /* optional (both or neither): the [K,M] operand is not `gout` itself but formed while it is staged,
       element (k, m) = (gout[k, m] > 0) ? outer_vec[k] * outer_w[m] : 0 — the gradient behind ReLU -> one-output Linear
       (the MP-DQN Q-head, core/networks.py:75-79): `gout` then holds the ReLU's OUTPUT, outer_vec the gradient of the
       one output per row, outer_w that layer's weight row.  Saves the launch that would materialise the product. */
    const float* outer_vec;               /* [K] */
    const float* outer_w;                 /* [M] */
} macjd_wgrad_io;

int64_t macjd_linear_wgrad_workspace_floats(int64_t K, int32_t M, int32_t N);
int macjd_linear_wgrad(const macjd_wgrad_io* io, void* hip_stream);
/* Up to MACJD_WGRAD_MAX_BATCH independent problems in ONE launch pair (all their (tile, chunk) partial products,
   then all their chunk sums): same arithmetic and summation order as macjd_linear_wgrad per problem.  The learner's
   backward pass defers its six weight gradients to one such call (they only feed .grad; nothing downstream waits). */
#define MACJD_WGRAD_MAX_BATCH 8
int macjd_linear_wgrad_many(const macjd_wgrad_io* ios, int32_t n, void* hip_stream);

/*
 * Gather `n_rows` whole rows (episodes) of up to 8 tensors in ONE launch: dst_k[i, :] = src_k[idx[i], :] as raw
 * bytes (row_bytes[k] each, multiples of 4).  Replaces the per-key index_select of
 * EpisodeReplayBuffer.sample (reference utils/replay_buffer.py:181-183).
 */
typedef struct macjd_gather_io {
    int32_t n_tensors, n_rows;
    const int64_t* idx;            /* [n_rows] device */
    const void* src[8]; void* dst[8]; int64_t row_bytes[8];
    int64_t dst_row_bytes[8];      /* destination row pitch (>= row_bytes, multiple of 4); 0 = row_bytes */
} macjd_gather_io;

int macjd_gather_rows(const macjd_gather_io* io, void* hip_stream);

/*
 * Input rows of the Q-head for the TAKEN action, out[n, :] = [h[n, 0..H-1], onehot_A(idx[n]), P[n]]  — the
 * F.one_hot / torch.cat sequence of RNNAgent.get_q_value_for_action (reference core/networks.py:160-174)
 * as one launch.  idx outside [0, A) gives an all-zero one-hot block (the caller validates indices when asked to).
 */
typedef struct macjd_qinput_io {
    int64_t n_rows;
    int32_t H, A;
    const float* h;   int64_t h_ld;      /* [n_rows, H] */
    const void* idx;  int32_t idx_elem_size, reserved;   /* [n_rows] int32 (4) or int64 (8), contiguous */
    const float* P;                      /* [n_rows] contiguous */
    float* out;       int64_t out_ld;    /* [n_rows, H + A + 1] */
} macjd_qinput_io;

int macjd_qhead_input(const macjd_qinput_io* io, void* hip_stream);

/*
 * Q-value of the TAKEN action for n rows — RNNAgent.get_q_value_for_action (reference core/networks.py:131-180: one_hot,
 * cat, Linear, ReLU, Linear), the only differentiable path of the learner's loss into the agent (core/qmix.py:161-184) —
 * as ONE launch instead of macjd_qhead_input + a library GEMM with bias / ReLU epilogue + macjd_rowdot:
 *   x[n, :]   = [h[n, 0..H-1], onehot_A(idx[n]), P[n]]                   (written: the backward's weight-gradient operand)
 *   act[n, u] = ReLU(sum_k W1[u, k] h[n, k] + b1[u] + W1[u, H + idx[n]] + W1[u, H + A] P[n])   (written: saved for backward)
 *   q[n]      = sum_u w2[u] act[n, u] + b2
 * The H x H product runs on exact-f32 MFMA (32 rows per workgroup, wave w owns hidden units [16w, 16w+16), W1's h-columns
 * as register fragments); the one-hot / power columns of W1 are a gather from LDS.  H = 64 (= rows of W1), A <= 64,
 * h rows 16-byte aligned.  idx outside [0, A): empty one-hot block.  Differs from the three-launch form by the summation
 * order of the first layer (~1e-7 relative).
 */
typedef struct macjd_qtaken_io {
    int64_t n_rows;
    int32_t H, A;
    const float* h;   int64_t h_ld;                       /* [n_rows, H] */
    const void* idx;  int32_t idx_elem_size, reserved;    /* [n_rows] int32 (4) or int64 (8), contiguous */
    const float* P;                                       /* [n_rows] contiguous */
    const float* W1;  int64_t w1_ld;                      /* fc2_q_head.0.weight [H, H + A + 1] */
    const float* b1;  const float* w2;  const float* b2;  /* [H], fc2_q_head.2.weight [H], .bias [1] or NULL */
    float* x;         int64_t x_ld;                       /* [n_rows, H + A + 1] out (may be NULL) */
    float* act;       int64_t act_ld;                     /* [n_rows, H] out */
    float* q;                                             /* [n_rows] out */
} macjd_qtaken_io;

int macjd_qhead_taken_supported(int32_t H, int32_t A);   /* 1 / 0 */
int macjd_qhead_taken(const macjd_qtaken_io* io, void* hip_stream);

/*
 * LayerNorm forward over the last dimension (QMixer.state_norm, reference core/networks.py:215,270):
 *   mean = sum(x)/S, var = sum((x-mean)^2)/S (biased, two-pass), rstd = rsqrt(var + eps), y = (x-mean) rstd gamma + beta.
 * mean / rstd [M] are saved for torch's native_layer_norm_backward.  S <= 1024.
 */
typedef struct macjd_layernorm_io {
    int64_t M;
    int32_t S, reserved;
    float eps, reserved2;
    const float* x;  int64_t x_ld;
    const float* gamma;   /* [S] or NULL (= 1) */
    const float* beta;    /* [S] or NULL (= 0) */
    float* y;        int64_t y_ld;
    float* mean;          /* [M], optional */
    float* rstd;          /* [M], optional */
    float* xhat;     int64_t xhat_ld;   /* optional [M, S]: the normalised input (x - mean) rstd, before gamma / beta */
} macjd_layernorm_io;

int macjd_layernorm_forward(const macjd_layernorm_io* io, void* hip_stream);

/*
 * Gradients of a LayerNorm's gamma / beta when the normalised tensor feeds ONE Linear layer y = s W^T + b,
 * s = xhat gamma + beta (QMixer.state_norm -> the merged first hyper-network layer), from quantities the weight-
 * gradient pass already has:  dbeta[k] = sum_c gb[c] W[c,k],  dgamma[k] = sum_c W[c,k] G[c,k]  with gb = column sums
 * of the layer's output gradient and G = gout^T xhat ([C, K], one more split-K problem).  Replaces the [M, K]
 * input-gradient GEMM + the two reduction launches of native_layer_norm_backward.
 */
typedef struct macjd_lnparam_io {
    int32_t C, K;                          /* Linear: C outputs, K inputs (= LayerNorm width) */
    const float* W;   int64_t w_ld;        /* [C, K] */
    const float* G;   int64_t g_ld;        /* [C, K] = gout^T xhat */
    const float* gb;                       /* [C] */
    float* dgamma;                         /* [K] */
    float* dbeta;                          /* [K] */
} macjd_lnparam_io;

int macjd_layernorm_param_grad(const macjd_lnparam_io* io, void* hip_stream);
/* macjd_clip_adam_step_sample whose FIRST launch also evaluates the LayerNorm parameter gradients (this struct) — they are
   part of the clipped vector, at elements gamma_off / beta_off of io->grad (ln->dgamma / ln->dbeta must point there), so
   the squared-norm launch computes them in K extra workgroups and counts their squares itself; the other workgroups skip
   those two ranges.  One launch less behind the weight-gradient reduce.  io->partials: >= 256 + ln->K floats. */
int macjd_clip_adam_step_ln(const macjd_adam_io* io, const macjd_sampler_io* next, const macjd_lnparam_io* ln,
                            int64_t gamma_off, int64_t beta_off, void* hip_stream);

/*
 * Fused chain of up to three dense layers, y = act_n(W_n ... act_1(W_1 x + b_1) ... + b_n), float32 with
 * exact-f32 MFMA (v_mfma_f32_16x16x4_f32).  Replaces the separate Linear / activation launches of
 *   RNNAgent.actor            Linear-ReLU-Linear-ReLU-Linear-Sigmoid   (reference core/networks.py:54-61,127)
 *   fc1 + GRU input transform Linear-ReLU-Linear (W_ih x + b_ih)       (core/networks.py:100, GRUCell)
 * in the rollout (core/mac.py:168-187) and in the learner's time-parallel unroll (core/qmix.py:241-253).
 * W_l is torch.nn.Linear's [out, in] row-major weight.  Limits: dims[0] <= 256, hidden widths <= 128, last
 * width <= 384, ceil(width / 16) in {1,2,3,4,8,12,24}, one layer's LDS weight image (rows padded to a pitch of
 * 32 m + 2 floats) + the activation strips <= 160 KB of LDS; MACJD_EUNSUPPORTED otherwise.
 */
#define MACJD_ACT_NONE 0
#define MACJD_ACT_RELU 1
#define MACJD_ACT_SIGMOID 2
typedef struct macjd_mlp_io {
    int64_t n_rows;
    int32_t n_layers;        /* 1..3 */
    int32_t dims[4];         /* dims[0] = input width, dims[l+1] = output width of layer l */
    int32_t act[3];          /* MACJD_ACT_* per layer */
    const float* W[3];       /* [dims[l+1], dims[l]] contiguous */
    const float* b[3];       /* [dims[l+1]] */
    const float* x;  int64_t x_ld;   /* [n_rows, dims[0]], row stride in elements (0 = one row broadcast to all) */
    float* y;        int64_t y_ld;   /* [n_rows, dims[n_layers]] */
} macjd_mlp_io;

/* One launch, no workspace: the weights are read in torch's own [out, in] layout (LDS-DMA into padded LDS rows). */
int macjd_mlp_forward(const macjd_mlp_io* io, void* hip_stream);
/* Two independent chains in ONE launch (workgroups are split between them): the actor chain and the fc1 -> W_ih chain
   of the rollout step over the same observation rows, or the eval and target actors of the learner. */
int macjd_mlp_forward_pair(const macjd_mlp_io* io0, const macjd_mlp_io* io1, void* hip_stream);

/*
 * Backward of "ReLU on the first Cr columns of a [M, Cr + Cp] matrix, pass the last Cp columns through, hand the
 * result out as column blocks" (the merged first layer of the mixer's hyper-networks, reference
 * core/networks.py:283-299): gout[m, c] = g_k[m, c - start_k] * (act[m, c] > 0) for the ReLU blocks and
 * g_pass[m, c - Cr] for the pass-through block, written in ONE launch (autograd would run cat + threshold_backward
 * + cat).  Up to 4 ReLU blocks; a NULL block gradient counts as zero.
 */
typedef struct macjd_splitrelu_bwd_io {
    int64_t M;
    int32_t n_blocks, Cp;             /* ReLU blocks (1..4); pass-through width (may be 0) */
    int32_t width[4];                 /* widths of the ReLU blocks, sum = Cr */
    const float* g[4];  int64_t g_ld[4];     /* block gradients [M, width[k]] */
    const float* g_pass; int64_t gp_ld;      /* [M, Cp] */
    const float* act;   int64_t act_ld;      /* [M, Cr] ReLU output saved by the forward */
    float* gout;        int64_t gout_ld;     /* [M, Cr + Cp] */
    const float* outer_w[4];                 /* NULL, or [width[k]]: block k's gradient is g[k][m] * outer_w[k][c] */
} macjd_splitrelu_bwd_io;

int macjd_splitrelu_backward(const macjd_splitrelu_bwd_io* io, void* hip_stream);

/*
 * y[n] = x[n, :] . w + b for a Linear layer with ONE output feature (the Q-head's second layer and the mixer's V
 * head, reference core/networks.py:78,247): one launch instead of a bias-broadcast copy + a 16 x 256-tile GEMM.
 * K <= 1024, K % 4 == 0.
 */
typedef struct macjd_rowdot_io {
    int64_t n_rows;
    int32_t K, reserved;
    const float* x;  int64_t x_ld;   /* [n_rows, K] row stride in elements */
    const float* w;                  /* [K] */
    const float* b;                  /* [1] or NULL */
    float* y;                        /* [n_rows] contiguous */
} macjd_rowdot_io;

int macjd_rowdot(const macjd_rowdot_io* io, void* hip_stream);

/*
 * Gate arithmetic of one GRU cell step for N rows (torch.nn.GRUCell after its two GEMMs, reference
 * core/networks.py:100-113): given gi = W_ih x + b_ih and gh = W_hh h + b_hh ([N,3H], gate order r, z, n),
 *   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (h - n) z + n,
 * written to h_out and, optionally, to a second destination (the batched runner's staging row) in the same launch
 * (replaces the fused-cell launch + the copy).  H must be a multiple of 4; h_out may alias h.  gi_ld = 0 broadcasts
 * ONE gi row to all N rows (the observation of every env / agent is the same static vector, so its input transform
 * is computed once per episode batch).
 */
typedef struct macjd_grugates_io {
    int64_t n_rows;
    int32_t H, reserved;
    const float* gi;  int64_t gi_ld;
    const float* gh;  int64_t gh_ld;
    const float* h;   int64_t h_ld;
    float* h_out;     int64_t ho_ld;
    float* h_out2;    int64_t ho2_ld;   /* optional */
} macjd_grugates_io;

int macjd_gru_gates(const macjd_grugates_io* io, void* hip_stream);

/*
 * The whole QMix mixer as ONE launch each way (reference core/networks.py:250-315, QMixer.forward):
 *   s~ = LayerNorm(s)                                                                     networks.py:283
 *   [h_w1 | h_wf | h_V | b1_raw] = s~ W_first^T + b_first, ReLU on the first 2 Hh + Em columns   (first layers of
 *                                   hyper_w_1, hyper_w_final, V and the one-layer hyper_b_1, merged: 2 Hh + 2 Em rows)
 *   w1_raw = h_w1 W2^T + b2 [J Em]    wf_raw = h_wf Wf2^T + bf2 [Em]    v_raw = h_V . wV2 + bV2
 *   y = ELU(q . clamp(w1_raw,0,5) + clamp(b1_raw,-5,5)) . clamp(wf_raw,0,5) + clamp(v_raw,-5,5)
 * on the matrix cores in exact float32 (v_mfma_f32_16x16x4_f32).  A workgroup owns 16 rows; its four waves split the
 * OUTPUT columns of every layer (M = batch x steps is only a few thousand rows: 16-row workgroups fill the chip, and a
 * row's whole chain stays inside one workgroup).  Weights are read as MFMA operand fragments straight from L2 — every
 * fragment of the launch is requested before the first arithmetic instruction, so the chain pays one memory latency —
 * activations pass between the layers through LDS, and the hyper-network outputs w1_raw / wf_raw are consumed from the
 * accumulators: they are never written to memory (the unfused form materialises w1_raw [M, J Em] in HBM and
 * contracts it in a second kernel).  Replaces 7 launches of the mixer forward (LayerNorm, 3 library GEMMs, ReLU,
 * row-dot, tail) by one, for the eval and the target mixer alike.
 *
 * Training (save != 0): the forward also stores what the weight-gradient products need — s~, xhat = (s - mean) rstd
 * and the first-layer output `act` [M, 2 Hh + 2 Em] (post-ReLU blocks, then b1_raw).
 * Backward (macjd_mixer_fused_backward): re-derives w1_raw / wf_raw / v_raw from `act` (same instruction sequence,
 * bit-identical), then writes dL/dq [M,J], the output gradients of the second layers (g_w1raw [M,J Em], g_wfraw [M,Em],
 * g_v [M]) and — through the transposed second-layer weights and the ReLU masks — the output gradient of the merged
 * first layer `gout1` [M, 2 Hh + 2 Em].  No gradient flows to the state.  The weight / bias / LayerNorm-parameter
 * gradients are split-K products of these matrices (macjd_linear_wgrad_many, macjd_layernorm_param_grad).
 *
 * Supported: hyper_hidden_dim Hh = 128, mixing_embed_dim Em = 64 (the reference's sizes), n_agents J in {2, 3, 6, 12},
 * state_dim S <= 16 J (the shipped 2j/2r, 3j/4r, 6j/8r and 12j/16r scenarios; J = 12 runs both layers in passes); everything else returns
 * MACJD_EUNSUPPORTED and the caller keeps the unfused kernels.
 */
typedef struct macjd_mixerf_io {
    int64_t M;                 /* rows */
    int32_t J, S, Hh, Em;      /* agents, state_dim, hyper_hidden_dim, mixing_embed_dim */
    int32_t save, reserved;    /* forward: != 0 stores sn / xhat / act */
    float ln_eps;  float reserved_f;
    const float* s;   int64_t s_ld;      /* [M,S] state rows */
    const float* q;            /* [M,J] agent Q-values (contiguous) */
    const float* ln_w; const float* ln_b;      /* state_norm.{weight,bias} [S] */
    const float* W1; const float* b1;          /* merged first layer [2Hh+2Em, S] row-major (row stride S), bias [2Hh+2Em]:
                                                  rows = hyper_w_1.0 | hyper_w_final.0 | V.0 | hyper_b_1 */
    const float* W2; const float* b2;          /* hyper_w_1.2     [J Em, Hh], [J Em] */
    const float* Wf2; const float* bf2;        /* hyper_w_final.2 [Em, Hh],  [Em]   */
    const float* wV2; const float* bV2;        /* V.2             [1, Em],   [1]    */
    float* y;                  /* [M] Q_tot (forward out) */
    float* sn; float* xhat;    /* [M,S] contiguous (forward out when save; sn unused by backward) */
    float* act;                /* [M, 2Hh+2Em] contiguous (forward out when save, backward in) */
    const float* gy;           /* [M] dL/dy (backward in) */
    float* gq;                 /* [M,J]        (backward outs) */
    float* gout1;              /* [M, 2Hh+2Em] */
    float* g_w1raw;            /* [M, J Em]    */
    float* g_wfraw;            /* [M, Em]      */
    float* g_v;                /* [M]          */
} macjd_mixerf_io;

int macjd_mixer_fused_supported(int32_t J, int32_t S, int32_t Hh, int32_t Em);   /* 1 / 0 */
int macjd_mixer_fused_forward(const macjd_mixerf_io* io, void* hip_stream);
/* The eval and the target mixer of one learner update (reference core/qmix.py:151 and :187: the same module class on
 * two parameter sets and two Q inputs, same states) as ONE grid: `saved` (save = 1: the eval mixer, whose activations the
 * backward needs) and `plain` (save = 0) must agree in J / S / M.  Results equal the two single launches bit for bit. */
int macjd_mixer_fused_forward_pair(const macjd_mixerf_io* saved, const macjd_mixerf_io* plain, void* hip_stream);
int macjd_mixer_fused_backward(const macjd_mixerf_io* io, void* hip_stream);
/* macjd_mixer_fused_backward that forms dL/dy itself from the TD loss's inputs (macjd_td_loss's expression; td->y = this
   mixer's forward output, td->tq the target values, rows = td->B x td->gy_cols) and tot_m[0] = the batch's mask sum
   (macjd_td_mask_sum): io->gy is not read, td->gy is not written.  Takes the loss launch off the update's serial chain;
   bit-identical gradients.  td->stats != NULL: one extra workgroup of the same grid computes the loss's logged sums
   (stats[0..2] = loss, mean(y), mean(target) — core/qmix.py:194, 212-213; stats[3] is left alone), so that they need neither
   a launch nor a stream of their own. */
int macjd_mixer_fused_backward_td(const macjd_mixerf_io* io, const macjd_tdloss_io* td, const float* tot_m, void* hip_stream);

/*
 * The agent side of a WHOLE episode batch in one launch: for t = 0 .. T-1 and every (env, agent) row
 *   h_t = GRUCell(x, h_{t-1})  with the input transform gi = W_ih ReLU(fc1 obs) + b_ih given (it does not change
 *         within an episode: the observation is static, reference simulation/environment.py:479-522),
 *   Q(h_t, a, P_a) for all a (MP-DQN multi-pass Q-head, core/networks.py:131-180), availability mask, epsilon-greedy
 *   choice and gather of the chosen power (core/mac.py:59-166, utils/action_selectors.py:15-62)
 * i.e. T x (macjd_gru_gates + two GEMMs + macjd_qhead_select) of the step-by-step rollout.  Nothing the agent
 * computes depends on the environment's outputs when the observation is static, so the T steps of the agent need no
 * env step in between (the env steps of the whole episode then run as ONE launch too: macjd_env_step_many).
 *
 * A workgroup owns 16 environments (16 J rows = J MFMA row tiles, tile j = agent j) for the whole episode: W_hh and
 * the Q-head's h-columns live in registers as MFMA operand fragments (wave w owns hidden units [16w, 16w+16) of all
 * three gates), h_t ping-pongs between two LDS tiles, two barriers per step, no global synchronisation at all —
 * environments are independent.  Exact float32 (v_mfma_f32_16x16x4_f32); gate and Q-head expressions and the Philox
 * exploration draws are those of macjd_gru_gates / macjd_qhead_select (same (row, counter) keying), so the episode equals
 * the step-by-step rollout up to the summation order of the two matrix products.
 * Supported: H = 64, A in {5, 9, 17, 33}, any J >= 1 (J in {2, 3, 6} with A <= 17: one row tile per agent as described; every
 * other size: four tiles of 16 consecutive rows n = env * J + agent per workgroup — the arithmetic per row is the same);
 * MACJD_EUNSUPPORTED otherwise.
 */
typedef struct macjd_agent_episode_io {
    int64_t n_envs;            /* E */
    int32_t T, J, H, A;        /* steps, agents, rnn_hidden_dim, n_actions */
    int32_t greedy_only, reserved;
    const float* gi;    int64_t gi_ld;   /* [E*J, 3H] input transform incl. b_ih; gi_ld = 0: one row for all */
    const float* P_all; int64_t p_ld;    /* [E*J, A] actor output; p_ld = 0: one row for all */
    const float* h0;                     /* [E*J, H] initial hidden state or NULL = zeros (mac.init_hidden) */
    const float* w_hh;  const float* b_hh;           /* rnn.weight_hh [3H,H], rnn.bias_hh [3H] */
    const float* W1;    int64_t w1_ld;               /* fc2_q_head.0.weight [H, H+A+1] */
    const float* b1;    const float* w2; const float* b2;   /* fc2_q_head.0.bias [H], .2.weight [H], .2.bias [1] */
    const void* avail;  int32_t avail_elem_size, reserved2;  /* optional mask (static), int32 / int64 elements */
    int64_t av_se, av_sj, av_sa;
    const float* eps;          /* [T] exploration probability of every step (device memory) */
    uint64_t seed;
    const uint64_t* counter_base;   /* device scalar: step t draws with counter = *counter_base + t + 1 */
    /* outputs, time-major staging rows of the batched runner */
    float*   hidden;    /* [T(+1), E, J, H] contiguous: row t = post-update h_t */
    int32_t* T_out;     /* [T, E, J] chosen discrete action */
    float*   P_out;     /* [T, E, J] chosen power */
    float*   h_final;   /* [E*J, H] optional: h_{T-1} (mac.hidden_states) */
} macjd_agent_episode_io;

int macjd_agent_episode_supported(int32_t J, int32_t H, int32_t A);
int macjd_agent_episode(const macjd_agent_episode_io* io, void* hip_stream);

/*
 * Double-DQN target values straight from the unrolled hidden states (reference core/qmix.py:138-147): for every row n
 *   a* = argmax_a Q_eval(h_e[n], a, P_e[n,a])   (no availability mask there, qmix.py:141-142)
 *   out[n] = Q_target(h_t[n], a*, P_t[n,a*])
 * in ONE launch: both Q-head base products W1[:, :H] h + b1 on the matrix cores (exact f32), both all-action Q-heads,
 * arg-max and gather.  Replaces two library GEMMs + two macjd_qhead_select launches behind the learner's scan.
 * h_e / h_t [n, H] (the same tensor when the agent body is shared), P_e / P_t [n, A].  Supported: H = 64, A in {5, 9, 17}.
 */
typedef struct macjd_doubleq_io {
    int64_t n_rows;
    int32_t H, A;
    const float* h_e; int64_t he_ld;  const float* h_t; int64_t ht_ld;
    const float* P_e; int64_t pe_ld;  const float* P_t; int64_t pt_ld;
    const float* W1_e; int64_t w1e_ld; const float* b1_e; const float* w2_e; const float* b2_e;   /* eval fc2_q_head */
    const float* W1_t; int64_t w1t_ld; const float* b1_t; const float* w2_t; const float* b2_t;   /* target fc2_q_head */
    float* out;              /* [n] */
    int64_t* argmax_out;     /* optional [n] */
    /* optional row map of P_e / P_t: p_group > 0 -> row n reads P row (n / p_group) * p_inner + n % p_inner (one actor
       row per SEQUENCE (b, j) for rows n = (b, t, j): p_group = T * J, p_inner = J) */
    int64_t p_group, p_inner;
} macjd_doubleq_io;

int macjd_qhead_double_q_supported(int32_t H, int32_t A);
int macjd_qhead_double_q(const macjd_doubleq_io* io, void* hip_stream);
/* macjd_qhead_taken and macjd_qhead_double_q of one learner update (reference core/qmix.py:138-147 and :161-184: both read
 * the same unrolled hidden states, neither reads the other's result) as ONE grid; same A in both.  Results equal the two
 * single launches bit for bit. */
int macjd_qheads_pair(const macjd_qtaken_io* taken, const macjd_doubleq_io* dq, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MACJD_NETS_H */
