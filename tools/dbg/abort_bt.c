/* LD_PRELOAD helper: print the native backtrace of whichever thread calls abort() or raises SIGSEGV / SIGABRT (python's
 * faulthandler shows Python frames only).  abort() itself is interposed (calls from shared libraries resolve here first), so
 * the trace is printed before any signal handling; backtrace() is warmed up at load time so that the handlers allocate nothing.
 *   gcc -shared -fPIC -O1 -o abort_bt.so abort_bt.c -ldl ;  LD_PRELOAD=tools/dbg/abort_bt.so python -m pytest ...               */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void dump(const char* why) {
    void* frames[128];
    const char msg[] = "\n==== abort_bt: native backtrace (";
    (void)!write(2, msg, sizeof(msg) - 1);
    (void)!write(2, why, strlen(why));
    (void)!write(2, ") ====\n", 7);
    int n = backtrace(frames, 128);
    backtrace_symbols_fd(frames, n, 2);
    (void)!write(2, "==== end ====\n", 14);
}
static void handler(int sig) {
    dump(sig == SIGSEGV ? "SIGSEGV" : "SIGABRT");
    signal(sig, SIG_DFL);
    raise(sig);
}
void abort(void) {
    static void (*real)(void) = 0;
    dump("abort() called");
    if (!real) real = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    if (real) real();
    _exit(134);
}
__attribute__((constructor)) static void install(void) {
    void* warm[4];
    backtrace(warm, 4); /* loads libgcc now, not inside a handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = handler;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
}
