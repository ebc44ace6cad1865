// Shared thread-local error slot behind macjd_last_error() (defined in macjd_env.hip).
#pragma once
namespace macjd {
int set_err(int code, const char* fmt, const char* a = "");
}
