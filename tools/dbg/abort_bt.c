/* LD_PRELOAD helper: print the native backtrace of whichever thread raises SIGABRT / SIGSEGV (python's faulthandler shows Python
 * frames only).   gcc -shared -fPIC -O1 -o abort_bt.so abort_bt.c ;  LD_PRELOAD=tools/dbg/abort_bt.so python -m pytest ...      */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
    void* frames[96];
    const char msg[] = "\n==== abort_bt: native backtrace of the signalling thread ====\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = handler;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
}
