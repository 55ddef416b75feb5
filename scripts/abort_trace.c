/* LD_PRELOAD helper for GPU-box debugging: prints a native backtrace of the thread that raised SIGABRT / SIGSEGV
 * (gcc -shared -fPIC -o /tmp/abort_trace.so scripts/abort_trace.c).  Not part of the product. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static int out_fd = 2;       /* the stderr the process was started with: pytest's fd capture redirects fd 2 later */

/* what the process wrote last to fd 1 / fd 2 (under pytest's fd capture: unlinked temporary files, still readable through /proc) */
static void dump_tail(const char *path, const char *title)
{
    static char buf[6144];
    int fd = open(path, O_RDONLY);
    if (fd < 0) return;
    off_t end = lseek(fd, 0, SEEK_END);
    if (end > 0) {
        off_t from = end > (off_t)sizeof buf ? end - (off_t)sizeof buf : 0;
        ssize_t n = pread(fd, buf, sizeof buf, from);
        if (n > 0 && write(out_fd, title, strlen(title)) >= 0 && write(out_fd, buf, (size_t)n) < 0) _exit(99);
    }
    close(fd);
}

static void on_signal(int sig)
{
    void *frames[64];
    const char *msg = sig == SIGABRT ? "\n[abort_trace] SIGABRT, native backtrace:\n" : "\n[abort_trace] SIGSEGV, native backtrace:\n";
    if (write(out_fd, msg, strlen(msg)) < 0) _exit(99);
    backtrace_symbols_fd(frames, backtrace(frames, 64), out_fd);
    dump_tail("/proc/self/fd/2", "\n[abort_trace] tail of fd 2:\n");
    dump_tail("/proc/self/fd/1", "\n[abort_trace] tail of fd 1:\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    out_fd = dup(2);
    if (out_fd < 0) out_fd = 2;
    signal(SIGABRT, on_signal);
    signal(SIGSEGV, on_signal);
}
