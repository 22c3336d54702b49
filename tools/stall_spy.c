/* tools/stall_spy.c: a signal handler that prints the interrupted thread's user-space backtrace (tools/stall_hunt.py, STALL_SPY=1) */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void spy_handler(int sig, siginfo_t *si, void *ctx)
{
    void *pc[48];
    (void)sig; (void)si; (void)ctx;
    int n = backtrace(pc, 48);
    static const char head[] = "#### user-space backtrace of the main thread:\n";
    if (write(2, head, sizeof head - 1) < 0) return;
    backtrace_symbols_fd(pc, n, 2);
}

int spy_install(int sig)
{
    struct sigaction sa;
    void *warm[4];
    backtrace(warm, 4);                 /* (loads the unwinder now, not inside the handler) */
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = spy_handler;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    return sigaction(sig, &sa, 0);
}
