// Error state and library identification for the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ried_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ried_last_error(void) { return g_err; }
extern "C" int ried_abi_version(void) { return 1; }

// Number of compute units of the current device (used by hosts to size split-K); 0 on failure.
extern "C" int ried_device_cus(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// Native backtrace on a fatal signal.  Python's faulthandler prints the interpreter's frames of the thread that
// holds the GIL; a host abort raised inside the HIP runtime, RCCL or a C++ destructor (possibly on a thread
// that never ran Python) shows nothing there.  The handler below writes the native frames of the thread that
// received the signal to stderr and to an optional log file, then hands the signal to whatever handler was
// installed before it (faulthandler's, or the default action).  Only async-signal-safe calls are used after
// installation: write, backtrace (glibc unwinder, primed once at installation), backtrace_symbols_fd, sigaction,
// raise.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static const int kCrashSignals[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL};
static const int kNumCrashSignals = sizeof(kCrashSignals) / sizeof(kCrashSignals[0]);
static struct sigaction g_prev_action[kNumCrashSignals];
static int g_crash_fds[2] = {2, -1};
static volatile sig_atomic_t g_crash_installed = 0;

static void crash_write(int fd, const char* s) {
  if (fd >= 0) (void)!write(fd, s, strlen(s));
}

static void crash_write_num(int fd, long v) {
  char buf[24];
  int n = 0;
  if (v == 0) buf[n++] = '0';
  bool neg = v < 0;
  if (neg) v = -v;
  while (v > 0 && n < 22) { buf[n++] = (char)('0' + v % 10); v /= 10; }
  if (neg) buf[n++] = '-';
  for (int i = 0; i < n / 2; ++i) { char t = buf[i]; buf[i] = buf[n - 1 - i]; buf[n - 1 - i] = t; }
  if (fd >= 0) (void)!write(fd, buf, n);
}

static void crash_handler(int sig) {
  void* frames[96];
  int n = backtrace(frames, 96);
  long tid = (long)syscall(SYS_gettid), pid = (long)getpid();
  for (int k = 0; k < 2; ++k) {
    int fd = g_crash_fds[k];
    if (fd < 0) continue;
    crash_write(fd, "\n==== libried_gan: fatal signal ");
    crash_write_num(fd, sig);
    crash_write(fd, tid == pid ? " on the main thread (tid " : " on a secondary thread (tid ");
    crash_write_num(fd, tid);
    crash_write(fd, ", pid ");
    crash_write_num(fd, pid);
    crash_write(fd, "); native frames, innermost first ====\n");
    backtrace_symbols_fd(frames, n, fd);
    crash_write(fd, "==== end of native frames ====\n");
  }
  // hand over to the previous handler: put it back and raise again (a handler that returns from a synchronous
  // fault re-executes the faulting instruction, so the same chain works for SIGSEGV)
  for (int i = 0; i < kNumCrashSignals; ++i)
    if (kCrashSignals[i] == sig) sigaction(sig, &g_prev_action[i], nullptr);
  raise(sig);
}

// Installs the handler for SIGABRT / SIGSEGV / SIGBUS / SIGFPE / SIGILL (idempotent).  log_path_host: NULL, or a
// file that receives a copy of the backtrace (opened for append now, so the handler opens nothing);
// extra_fd: -1, or a descriptor to write to instead of 2 (a test runner that redirects descriptor 2 per test
// passes its saved copy of the real stderr).  No counterpart in the reference; diagnostics only.
extern "C" int ried_install_crash_backtrace(const char* log_path_host, int extra_fd) {
  if (g_crash_installed) return 0;
  void* prime[4];
  backtrace(prime, 4);      // loads the unwinder now: its first call allocates
  if (extra_fd >= 0) g_crash_fds[0] = extra_fd;
  if (log_path_host != nullptr && log_path_host[0] != 0) {
    int fd = open(log_path_host, O_WRONLY | O_CREAT | O_APPEND | O_CLOEXEC, 0644);
    if (fd < 0) {
      ried_set_error("ried_install_crash_backtrace: cannot open %s", log_path_host);
      return 1;
    }
    g_crash_fds[1] = fd;
  }
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = crash_handler;
  sigemptyset(&sa.sa_mask);
  sa.sa_flags = SA_NODEFER | SA_ONSTACK;
  for (int i = 0; i < kNumCrashSignals; ++i)
    if (sigaction(kCrashSignals[i], &sa, &g_prev_action[i]) != 0) {
      ried_set_error("ried_install_crash_backtrace: sigaction(%d) failed", kCrashSignals[i]);
      return 1;
    }
  g_crash_installed = 1;
  return 0;
}
