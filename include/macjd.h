/*
 * macjd.h — C-ABI of libmacjd_hip.so, the MI355X (gfx950) native library behind the
 * batched radar-jamming environment step.
 *
 * The reference (mfathulkr/MA-CJD-Cooperative-Jamming-Decision-Making-via-MARL) is pure Python
 * and has no FFI layer; its boundary for this path is the object protocol that main.py drives
 * (main.py:137-162,193).  Each entry point below names the reference interface it replaces.
 * Everything is `extern "C"`, plain pointers and sizes; no torch / HIP types in the signatures
 * (streams travel as `void*` holding a hipStream_t).
 *
 * Conventions
 *   - all data buffers are caller-owned DEVICE pointers (e.g. torch tensors' data_ptr());
 *   - entry points never allocate, free or synchronise on the hot path (graph-capture safe),
 *     they enqueue on the given stream and return;
 *   - return 0 on success, a negative MACJD_E* code otherwise; the message is available from
 *     macjd_last_error() (thread-local);
 *   - env arrays are addressed with explicit ELEMENT strides so the caller picks the HBM layout:
 *     env-major [E,J] (se=J, sx=1) as the reference's runner produces it, or agent-major [J,E]
 *     (se=1, sx=E), which is the coalesced layout the batched runner uses.
 */
#ifndef MACJD_H
#define MACJD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MACJD_ABI_VERSION 3
#define MACJD_PE_ROWS(R, J) (6 * (R) + 3 * (J) + (J) * (R))

#define MACJD_OK          0
#define MACJD_EINVAL     -1  /* bad argument (shape, NULL, stride)            */
#define MACJD_ENOMEM     -2  /* device / host allocation failed               */
#define MACJD_EDEVICE    -3  /* HIP runtime error (no device, launch failure) */
#define MACJD_EUNSUPPORTED -4

#define MACJD_MAX_RADARS  32
#define MACJD_MAX_JAMMERS 32

/* jr_flags bits */
#define MACJD_JR_WEAK_DENOM 1u /* denominator was a Python float (d^2 <= 1e-9 branch of
                                  jammer.py:83): NumPy-2 weak promotion keeps the division in
                                  float32 when the power is float32 */

/* macjd_step_io.flags bits */
#define MACJD_STEP_ARITH_F64 1u /* P32 given, but do the power arithmetic in float64
                                   (NumPy-1.x value-based casting / python-float actions) */
/* Kernel choice (results are bit-identical).  Default: the (env x slot) kernel for the templated scenario
   sizes when n_envs < 2^17 (latency regime: shortest critical path), the one-lane-per-env kernel otherwise
   (throughput regime: fewest instructions per env).  These two bits force one variant (A/B + tests). */
#define MACJD_STEP_LANE_KERNEL 2u
#define MACJD_STEP_SLOT_KERNEL 4u

/*
 * Host-side description of one scenario: the static tables the scenario compiler derives from
 * the sim-config YAML.  Replaces the per-step recomputation inside
 * ElectromagneticEnvironment.step (simulation/environment.py:316-333: echo power, SNR without
 * jamming; core/radar.py:35-60; utils/math_utils.py:40-42 distances) — all of it is static
 * because nothing in step() moves an entity (environment.py:237-238 is a TODO).
 * All pointers are HOST pointers, copied by macjd_scenario_create.
 */
typedef struct macjd_scenario_desc {
    int32_t n_radars;       /* R, 1..MACJD_MAX_RADARS  */
    int32_t n_jammers;      /* J, 1..MACJD_MAX_JAMMERS */
    int32_t episode_limit;  /* environment.py:84,460   */
    int32_t reserved;
    double rp_min, rp_max;  /* environment.py:72-73,377 */
    /* Albersheim-style Pd constants, core/radar.py:67-82, evaluated on the host with the
       reference's own expressions so they are bit-identical: A = ln(0.62/prfa),
       c1 = 5 log10(m) / (6.2 + 4.54/sqrt(m) + 0.44), denB = 1.7 + 0.12 A */
    double pd_A, pd_c1, pd_denB;
    const double* radar_GaPs;   /* [R] Ga*Ps               environment.py:320-326, radar.py:35-60 */
    const double* radar_Pn;     /* [R] pn_watts            radar.py:19                            */
    const double* radar_D;      /* [R] anti_jamming_factor radar.py:28                            */
    const double* radar_pd_no;  /* [R] pd(snr_no_jam)      environment.py:385                     */
    const double* radar_snr_no; /* [R] max(0,Ga*Ps/Pn)     environment.py:326-327                 */
    const double* radar_rd_pen; /* [R] clip(-threat, rd_min, rd_max)  environment.py:362-365     */
    const double* radar_gr;     /* [R] receive gain (linear) = get_antenna_gain, radar.py:84-85  */
    const double* jam_pmin;     /* [J] power_min           environment.py:185                     */
    const double* jam_pmax;     /* [J] power_max           environment.py:184                     */
    const double* jam_gj;       /* [J] gain (linear)       jammer.py:44                           */
    const double* jr_denom;     /* [J*R] max(1e-9,d^2)*loss*latm*max(1e-9,bj), jammer.py:83-88;
                                   NEGATIVE when d <= 1e-6 (action ignored, environment.py:284)  */
    const uint8_t* jr_flags;    /* [J*R] MACJD_JR_* */
} macjd_scenario_desc;

typedef struct macjd_scenario macjd_scenario; /* opaque, owns the device copy of the tables */

/*
 * One env-step call.  Replaces ElectromagneticEnvironment.step (simulation/environment.py:221-477)
 * for E environments at once.  Device pointers; NULL where marked optional.
 */
typedef struct macjd_step_io {
    int64_t n_envs;      /* E */
    int64_t env_offset;  /* global index of env 0 (rank shard offset); keys the in-kernel RNG */
    uint64_t seed;       /* Philox key when u == NULL */
    uint32_t flags;      /* MACJD_STEP_* */
    uint32_t reserved;

    /* actions: T = discrete index (environment.py:249-268), P = normalised power (:251,271) */
    const int32_t* T;    int64_t T_se, T_sx;   /* [E,J] element strides (env, agent) */
    const float*   P32;  /* exactly one of P32 / P64 is non-NULL */
    const double*  P64;  int64_t P_se, P_sx;

    /* uniforms replacing np.random.rand() (environment.py:341,430): slot r<R = radar r's
       detection draw, slot R+k = k-th valid deception action (jammer order).  NULL → generated
       in-kernel by Philox4x32-10: one block serves FOUR slots,
         key     = (seed[31:0], seed[63:32] ^ genv[63:32]),            genv = env_offset + e
         counter = (genv[31:0], episode[e], step_before, slot >> 2),
         u       = (word[slot & 3] + 0.5) * 2^-32   (never 0, never 1),
       so every (env, episode, step, slot) has its own value whatever the batch size, the sharding
       over GPUs or the launch geometry. */
    const double* u;     int64_t u_se, u_sx;   /* [E,R+J] */
    /* [E] episode index of each env, optional (NULL = 0): the reference draws fresh np.random.rand()
       values in every episode; macjd_env_reset advances this counter (device memory, so a replayed
       HIP graph advances it too) and the in-kernel generator keys on it.  Unused when u != NULL. */
    const int32_t* episode;

    /* state, read-modify-write */
    uint8_t* track;      int64_t k_se, k_sx;   /* [E,R] 1 = TRACK, 0 = SEARCH (radar.py:90-119) */
    int32_t* step;       /* [E] _step_count (environment.py:235) */

    /* outputs */
    float*   reward;     /* [E]   r_d + r_p + r_j (environment.py:457), may be NULL */
    float*   r_dpj;      /* [E,3] contiguous (r_d, r_p, r_j), may be NULL            */
    uint8_t* terminated; /* [E]   step >= episode_limit (environment.py:460), may be NULL */
    float*   pd;         int64_t pd_se, pd_sx; /* [E,R] info['radar_pds'], may be NULL       */
    float*   snr_with;   int64_t sw_se, sw_sx; /* [E,R] info['snr_with_jamming'], may be NULL */
    /* optional float64 diagnostics (single-env facade + parity tests) */
    double*  out64;      /* [E,4] contiguous (reward, r_d, r_p, r_j) in float64, may be NULL  */
    double*  pd64;       /* [E,R] contiguous, may be NULL */
    double*  snr64;      /* [E,R] contiguous, may be NULL */
    double*  prj64;      /* [E,J] contiguous: received jamming power of jammer j's action if it
                            was recorded in info['jammer_actions'] (environment.py:288-295),
                            else -1; may be NULL */
    /* Per-env scenario tables, optional (radar / jammer positions, threat levels ... randomised per
       env — the mode in which the static scenario parameters carry real HBM bytes).  pe_tables is float64
       [MACJD_PE_ROWS(R,J), pe_stride] row-major: row t of env e at pe_tables[t * pe_stride + e] (consecutive envs are
       consecutive lanes: coalesced).  Row order, same quantities as macjd_scenario_desc:
         GaPs[R] | Pn[R] | D[R] | pd_no[R] | rd_pen[R] | gr[R] | pmin[J] | pmax[J] | gj[J] | denom[J*R] (j-major)
       pe_flags is uint8 [J*R, pe_stride] (MACJD_JR_*).  NULL = every env uses the scenario handle's tables; the
       handle still supplies R, J, episode_limit, the r_p bounds and the Pd constants. */
    const double*  pe_tables;
    const uint8_t* pe_flags;
    int64_t        pe_stride;   /* >= n_envs */
    /* pe_tile > 0 selects the tiled (AoSoA) form of the same tables instead: envs are grouped in tiles of pe_tile
       consecutive envs and a tile's rows are contiguous — row t of env e at
       pe_tables[((e / pe_tile) * MACJD_PE_ROWS + t) * pe_tile + e % pe_tile] (flags alike with J*R rows) — so a
       workgroup that handles pe_tile consecutive envs reads ONE contiguous block per step instead of 6R+3J+JR
       streams pe_stride apart.  pe_stride is ignored then; the last tile is padded. */
    int32_t        pe_tile;
    int32_t        reserved_pe;
    float*   r_dpj_sum;  /* [E,3] contiguous, optional: (r_d, r_p, r_j) of this step are ADDED to it — the
                            per-episode sums behind run_info['avg_r_d'|'avg_r_p'|'avg_r_j']
                            (runners/episode_runner.py:88-90,141-143) without a separate launch */
} macjd_step_io;

/* library / device */
int         macjd_abi_version(void);
const char* macjd_last_error(void);
/* The library reads its MACJD_* environment switches (MACJD_ENV_REGULAR, MACJD_ENV_PD32, MACJD_GRU_SCAN) once, at the first
 * launch; call this after changing one of them inside a running process (tests, A/B timing). */
void        macjd_reload_options(void);
int         macjd_device_count(void);   /* number of HIP devices, <0 on error */

/* scenario handle: replaces ElectromagneticEnvironment.__init__/_initialize_entities
   (environment.py:35-206) as far as the device is concerned */
int  macjd_scenario_create(const macjd_scenario_desc* host_desc, macjd_scenario** out);
void macjd_scenario_destroy(macjd_scenario* s);
int  macjd_scenario_dims(const macjd_scenario* s, int32_t* n_radars, int32_t* n_jammers,
                         int32_t* episode_limit);
/* 1 when the scenario's tables are REGULAR: every value a division of the step can meet lies in [1e-30, 1e30] (or is an
   exact zero where that is harmless), noise powers are positive, the detection-probability argument cannot fall below
   -700.  The production lane kernel then divides by table values through reciprocals refined on the device when the
   scenario was created and drops the guards that cannot trigger — bit-identical results (tests), ~16 % fewer
   instructions.  Irregular scenarios run the IEEE-division form of the same kernel; MACJD_ENV_REGULAR=0 forces it. */
int  macjd_scenario_is_regular(const macjd_scenario* s);

/* replaces ElectromagneticEnvironment.reset (environment.py:208-219): all radars SEARCH,
   step counter 0.  mask (optional, [E] uint8) restricts the reset to envs with mask != 0.
   episode (optional, [E] int32): the reset envs' episode index is incremented (see
   macjd_step_io.episode). */
int macjd_env_reset(const macjd_scenario* s, int64_t n_envs, uint8_t* track, int64_t k_se,
                    int64_t k_sx, int32_t* step, const uint8_t* mask, int32_t* episode,
                    void* hip_stream);

/* replaces ElectromagneticEnvironment.step (environment.py:221-477) */
int macjd_env_step(const macjd_scenario* s, const macjd_step_io* io, void* hip_stream);

/* T consecutive steps of all E environments in ONE launch, given the actions of all T steps (time-major: step t of
   env e at offset t * t_stride + e * se + k * sx of T / P32; outputs reward / terminated / r_dpj likewise at
   t * n_envs + e, pd / snr_with are not written).  Legal because a step's outcome depends on the environment's past
   only through the step counter: the FSM's next state equals `detected` whatever the previous state (core/radar.py:
   102-117), nothing else is carried over (environment.py:221-477).  So when the actions do not depend on the env's
   outputs — the observation is static, environment.py:479-522 — the T x E env-steps of an episode batch are
   independent work items: virtual env v = t * E + e runs (env e, step step[e] + t) in the streaming lane kernel.
   `track` and `step` end as after the last step; io->r_dpj_sum gets the sums over the T steps.  Production
   configuration only (Philox uniforms, float32 actions); io->u, P64 and the float64 diagnostics must be NULL. */
int macjd_env_step_many(const macjd_scenario* s, const macjd_step_io* io, int32_t n_steps, int64_t t_stride,
                        void* hip_stream);

/* timing helper for bench.py: `iters` back-to-back env_step launches, replayed from one HIP graph on a private
   stream after the work queued on `hip_stream` has finished (as the benchmark's rollout replays them, so the
   host's enqueue rate does not enter; issued directly on `hip_stream` if the capture is refused), bracketed by
   HIP events recorded on the stream the launches run on; waits and returns the average milliseconds per
   launch in *ms_per_launch. */
int macjd_env_step_timed(const macjd_scenario* s, const macjd_step_io* io, int iters,
                         void* hip_stream, float* ms_per_launch);

/* the same for the main kernel of macjd_env_step_many (n_steps x n_envs work items per launch; the counter-advance
   launch is not part of the replayed graph, so every replay does identical work) */
int macjd_env_step_many_timed(const macjd_scenario* s, const macjd_step_io* io, int32_t n_steps, int64_t t_stride,
                              int iters, void* hip_stream, float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* MACJD_H */
