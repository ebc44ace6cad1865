// Shared thread-local error slot behind macjd_last_error() (defined in macjd_env.hip).
#pragma once
namespace macjd {
int set_err(int code, const char* fmt, const char* a = "");

// Process-wide switches of the launches, read from the environment ONCE (macjd_reload_options() re-reads them: tests
// and A/B runs that change a switch inside one process).  Defined in macjd_env.hip.
struct EnvOptions {
    bool regular;      // MACJD_ENV_REGULAR=0: env lane kernel keeps IEEE divisions and every guard
    bool pd32;         // MACJD_ENV_PD32=0: all-float64 detection probabilities in the production env variant
    bool gru_ksplit;   // MACJD_GRU_SCAN=ksplit: the K-split GRU scan at H = 64
};
const EnvOptions& env_options();
}
