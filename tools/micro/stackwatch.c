// Debug helper (not part of the product): a watcher thread signals the main thread every `period_us` while armed and the handler
// records the native backtrace; stackwatch_dump() prints every run of >= `min_run` consecutive samples with identical frames,
// i.e. where the host thread sat during a long native call.  gcc -O1 -shared -fPIC -o stackwatch.so stackwatch.c -lpthread
#define _GNU_SOURCE
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define MAXS 8192
#define DEPTH 40
static pthread_t main_thread, watcher;
static volatile int armed = 0, running = 0;
static int period_us = 5000;
static void *frames[MAXS][DEPTH];
static int depth[MAXS];
static volatile int ns = 0;

static void handler(int sig)
{
    (void)sig;
    if (ns < MAXS) { depth[ns] = backtrace(frames[ns], DEPTH); ns++; }
}

static void *watch(void *arg)
{
    (void)arg;
    while (running) {
        usleep(period_us);
        if (armed) pthread_kill(main_thread, SIGUSR2);
    }
    return 0;
}

int stackwatch_start(int period)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, 0);
    main_thread = pthread_self();
    period_us = period;
    running = 1;
    void *warm[4];
    backtrace(warm, 4);                    // loads libgcc outside the handler
    return pthread_create(&watcher, 0, watch, 0);
}
void stackwatch_arm(int on) { armed = on; }
int stackwatch_samples(void) { return ns; }
void stackwatch_dump(int min_run, int skip)
{   // frames 0..skip-1 are the handler / signal trampoline
    int i = 0;
    while (i < ns) {
        int j = i + 1;
        while (j < ns && depth[j] == depth[i] && !memcmp(frames[j] + skip, frames[i] + skip, (depth[i] - skip) * sizeof(void *))) j++;
        if (j - i >= min_run) {
            fprintf(stderr, "---- %d consecutive identical samples (from sample %d of %d) ----\n", j - i, i, ns);
            fflush(stderr);
            backtrace_symbols_fd(frames[i], depth[i], 2);
        }
        i = j;
    }
    ns = 0;
}
