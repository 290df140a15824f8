// bamqualcheck — command-line front end; everything lives in libbamqc_gpu.so (bqc_main, bqc_main_multi).
//
// One GPU: the work is done by a child process; this process leaves with the child's exit status as soon as the child reports
// that the run is complete (output file written and closed, every message flushed).  What the child still does after that
// point — handing ~3 GB of page-locked buffers and the GPU context back to the kernel — takes 0.2 s that nobody has to wait
// for (a run started during that time pays for it in its own start-up: bench.py's `e2e.back_to_back` prices that).
// SIGINT / SIGTERM sent to this process reach the worker, and a front end that is killed takes its worker with it.
// BQC_NO_FORK=1 runs everything in this process (profilers, debuggers).
//
// `--gpus N` (N > 1): one worker process per GPU, forked here before anything touches a GPU; each takes its byte range of the BAM
// file, one RCCL reduce sums the state vectors (bamqc_amd/host/multi_gpu.cpp).  This process waits for all of them.
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../include/bamqc_host.h"

static pid_t g_child = 0;
static void forward_signal(int sig) { if (g_child > 0) kill(g_child, sig); }

int main(int argc, const char** argv)
{
    setenv("BQC_FAST_EXIT", "1", 0); // a finished run exits without the HIP runtime's static destructors (see driver.cpp)
    // --gpus N / --gpus=N: taken out of the arguments here (the reference's options are bqc_main's)
    int n_gpus = 0;
    bool plain = false; // -h / --help / --version: answered once, by this process
    std::vector<const char*> rest;
    for (int i = 0; i < argc; ++i) {
        if (i > 0 && strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) { n_gpus = atoi(argv[++i]); if (n_gpus < 1) n_gpus = -1; continue; }
        if (i > 0 && strncmp(argv[i], "--gpus=", 7) == 0) { n_gpus = atoi(argv[i] + 7); if (n_gpus < 1) n_gpus = -1; continue; }
        if (i > 0 && (strcmp(argv[i], "-h") == 0 || strcmp(argv[i], "--help") == 0 || strcmp(argv[i], "--version") == 0)) plain = true;
        rest.push_back(argv[i]);
    }
    if (n_gpus < 0) { fprintf(stderr, "bamqualcheck: --gpus wants a positive number\n"); return 1; }
    const char* force = getenv("BQC_GPUS_FORCE"); // (1: also --gpus 1 goes through the multi-GPU path: its RCCL set-up and reduce on one card)
    if (!plain && (n_gpus > 1 || (n_gpus == 1 && force && force[0] == '1'))) return bqc_main_multi((int)rest.size(), rest.data(), n_gpus);
    argc = (int)rest.size();
    argv = rest.data();
    const char* nf = getenv("BQC_NO_FORK");
    int fds[2];
    if (plain || (nf && nf[0] == '1') || pipe(fds) != 0) return bqc_main(argc, argv);
    const pid_t pid = fork(); // (before anything touches the GPU)
    if (pid < 0) { close(fds[0]); close(fds[1]); return bqc_main(argc, argv); }
    if (pid == 0) {
        (void)prctl(PR_SET_PDEATHSIG, SIGTERM); // (cleared by driver.cpp just before it reports the run complete)
        close(fds[0]);
        char fdname[16];
        snprintf(fdname, sizeof fdname, "%d", fds[1]);
        setenv("BQC_DONE_FD", fdname, 1); // driver.cpp reports the status there when the run is complete
        const int rc = bqc_main(argc, argv);
        fflush(stdout); fflush(stderr);
        const unsigned char st = (unsigned char)rc;
        if (write(fds[1], &st, 1) != 1) return rc;
        return rc;
    }
    close(fds[1]);
    g_child = pid;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = forward_signal;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGINT, &sa, nullptr);
    sigaction(SIGTERM, &sa, nullptr);
    unsigned char st = 1;
    const ssize_t got = read(fds[0], &st, 1);
    if (got == 1) return st; // the run is complete; the child finishes its teardown on its own
    int ws = 0;               // the child ended without reporting: its exit status (or signal) is the program's
    if (waitpid(pid, &ws, 0) < 0) return 1;
    return WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + (WIFSIGNALED(ws) ? WTERMSIG(ws) : 0);
}
