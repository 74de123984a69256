// LD_PRELOAD helper: native backtrace on SIGABRT / SIGSEGV (debugging aid; gcc -shared -fPIC -o abort_trace.so abort_trace.c)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void handler(int sig)
{
    void* frames[96];
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void)
{
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
