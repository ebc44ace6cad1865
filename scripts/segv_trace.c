// Diagnostic only (not product, not tests): LD_PRELOAD this to get the NATIVE backtrace of a host SIGSEGV / SIGBUS /
// SIGABRT — Python's faulthandler prints Python frames only.  Build: gcc -O1 -g -shared -fPIC -o libsegvtrace.so segv_trace.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

static FILE* g_out;   // MACJD_SEGV_TRACE_OUT (a test runner may have redirected fd 2) or stderr

static void maps_line_for(unsigned long a, const char* tag) {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return;
    char line[512];
    int found = 0;
    while (fgets(line, sizeof line, f)) {
        unsigned long lo, hi;
        if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && a >= lo && a < hi) {
            fprintf(g_out, "[segv_trace]   %s %#lx in: %s", tag, a, line);
            found = 1;
            break;
        }
    }
    if (!found) fprintf(g_out, "[segv_trace]   %s %#lx: NOT MAPPED\n", tag, a);
    fclose(f);
}

static struct sigaction g_old[32];

static void handler(int sig, siginfo_t* si, void* uc_) {
    ucontext_t* uc = (ucontext_t*)uc_;
    static volatile int entered = 0;
    if (entered++) { signal(sig, SIG_DFL); return; }   // never loop
    fprintf(g_out, "\n[segv_trace] signal %d code %d fault address %p\n", sig, si->si_code, si->si_addr);
    static const char* names[] = {"R8", "R9", "R10", "R11", "R12", "R13", "R14", "R15", "RDI", "RSI", "RBP", "RBX",
                                  "RDX", "RAX", "RCX", "RSP", "RIP"};
    for (int i = 0; i < 17; ++i) fprintf(g_out, "[segv_trace]   %s=%#llx\n", names[i], (unsigned long long)uc->uc_mcontext.gregs[i]);
    maps_line_for((unsigned long)si->si_addr, "fault");
    maps_line_for((unsigned long)uc->uc_mcontext.gregs[REG_RIP], "rip");
    void* frames[96];
    int n = backtrace(frames, 96);
    for (int i = 0; i < n; ++i) {
        Dl_info di;
        if (dladdr(frames[i], &di) && di.dli_fname)
            fprintf(g_out, "[segv_trace] #%d %p %s+%#lx (%s)\n", i, frames[i], di.dli_fname,
                    (unsigned long)((char*)frames[i] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
        else
            fprintf(g_out, "[segv_trace] #%d %p ?\n", i, frames[i]);
    }
    // bytes at RIP (to match the instruction in the stripped library)
    const unsigned char* ip = (const unsigned char*)uc->uc_mcontext.gregs[REG_RIP];
    fprintf(g_out, "[segv_trace] bytes at rip:");
    for (int i = 0; i < 16; ++i) fprintf(g_out, " %02x", ip[i]);
    fprintf(g_out, "\n");
    fflush(g_out);
    // hand over to whoever was installed before (Python's faulthandler prints the Python frames): restore it and return,
    // the faulting instruction runs again and raises the signal for that handler
    if (sig > 0 && sig < 32) sigaction(sig, &g_old[sig], NULL);
    else signal(sig, SIG_DFL);
}

// callable again (ctypes) after another handler was installed, so that this one runs FIRST and sees the original context
void segv_trace_install(void) {
    if (!g_out) {
        const char* path = getenv("MACJD_SEGV_TRACE_OUT");
        g_out = path ? fopen(path, "w") : NULL;
        if (!g_out) g_out = stderr;
    }
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, NULL);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    const int sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};
    for (int i = 0; i < 3; ++i) {
        struct sigaction prev;
        sigaction(sigs[i], &sa, &prev);
        if (prev.sa_sigaction != handler) g_old[sigs[i]] = prev;   // (a second install must not chain to itself)
    }
}

__attribute__((constructor)) static void install(void) { segv_trace_install(); }
