/*
 * macjd_oracle_mt.c — OpenMP driver over the scalar oracle.  TEST INFRASTRUCTURE ONLY
 * (bench.py's cpu_baseline leg times it on the GPU box's host cores; see macjd_oracle.c).
 * Environments are independent (no cross-env term anywhere in reference
 * simulation/environment.py:221-477), so the env range is simply cut into contiguous chunks.
 */
#include <stddef.h>
#include <stdint.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/macjd.h"

int macjd_oracle_env_step(const macjd_scenario_desc* d, const macjd_step_io* io, int32_t* draws_out);

static macjd_step_io slice(const macjd_scenario_desc* d, const macjd_step_io* io, int64_t lo, int64_t n) {
    macjd_step_io s = *io;
    const int R = d->n_radars, J = d->n_jammers;
    s.n_envs = n;
    s.env_offset = io->env_offset + lo;
    s.T += lo * io->T_se;
    if (s.P32) s.P32 += lo * io->P_se;
    if (s.P64) s.P64 += lo * io->P_se;
    if (s.u) s.u += lo * io->u_se;
    if (s.episode) s.episode += lo;
    s.track += lo * io->k_se;
    s.step += lo;
    if (s.reward) s.reward += lo;
    if (s.r_dpj) s.r_dpj += lo * 3;
    if (s.terminated) s.terminated += lo;
    if (s.pd) s.pd += lo * io->pd_se;
    if (s.snr_with) s.snr_with += lo * io->sw_se;
    if (s.out64) s.out64 += lo * 4;
    if (s.pd64) s.pd64 += lo * R;
    if (s.snr64) s.snr64 += lo * R;
    if (s.prj64) s.prj64 += lo * J;
    if (s.r_dpj_sum) s.r_dpj_sum += lo * 3;
    return s;
}

int macjd_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int macjd_oracle_env_step_mt(const macjd_scenario_desc* d, const macjd_step_io* io, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    const int64_t E = io->n_envs;
    int rc = 0;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int t = 0; t < n_threads; ++t) {
        int64_t lo = E * t / n_threads, hi = E * (t + 1) / n_threads;
        if (hi > lo) {
            macjd_step_io s = slice(d, io, lo, hi - lo);
            int r = macjd_oracle_env_step(d, &s, NULL);
            if (r != 0) {
#pragma omp critical
                rc = r;
            }
        }
    }
    return rc;
}

/*
 * CPU baseline timing for bench.py: n_steps consecutive env steps of all io->n_envs environments, timed INSIDE C on
 * the caller's pre-allocated buffers (no per-step Python, no allocation).  One parallel region for the whole run:
 * thread t owns a contiguous slice of the envs and steps it n_steps times (envs are independent, so no barrier is
 * needed between steps); the step counters are rewound every `episode_limit` steps like a reset would.  Returns the
 * wall time of the region in *seconds (max over threads by construction: the region ends when the last one does).
 */
int macjd_oracle_bench(const macjd_scenario_desc* d, const macjd_step_io* io, int n_threads, int n_steps, double* seconds) {
    if (n_threads < 1) n_threads = 1;
    if (n_steps < 1 || !seconds) return MACJD_EINVAL;
    const int64_t E = io->n_envs;
    int rc = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int t = 0; t < n_threads; ++t) {
        int64_t lo = E * t / n_threads, hi = E * (t + 1) / n_threads;
        if (hi > lo) {
            macjd_step_io s = slice(d, io, lo, hi - lo);
            for (int k = 0; k < n_steps; ++k) {
                if (k % d->episode_limit == 0)
                    for (int64_t e = 0; e < s.n_envs; ++e) s.step[e] = 0;
                int r = macjd_oracle_env_step(d, &s, NULL);
                if (r != 0) {
#pragma omp critical
                    rc = r;
                    break;
                }
            }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return rc;
}
