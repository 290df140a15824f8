// bamqualcheck — command-line front end; everything lives in libbamqc_gpu.so (bqc_main).
//
// The work is done by a child process; this process leaves with the child's exit status as soon as the child reports that
// the run is complete (output file written and closed, every message flushed).  What the child still does after that point —
// handing ~3 GB of page-locked buffers and the GPU context back to the kernel — takes 0.2 s that nobody has to wait for.
// BQC_NO_FORK=1 runs everything in this process (profilers, debuggers).
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <unistd.h>
#include "../include/bamqc_host.h"

int main(int argc, const char** argv)
{
    setenv("BQC_FAST_EXIT", "1", 0); // a finished run exits without the HIP runtime's static destructors (see driver.cpp)
    const char* nf = getenv("BQC_NO_FORK");
    int fds[2];
    if ((nf && nf[0] == '1') || pipe(fds) != 0) return bqc_main(argc, argv);
    const pid_t pid = fork(); // (before anything touches the GPU)
    if (pid < 0) { close(fds[0]); close(fds[1]); return bqc_main(argc, argv); }
    if (pid == 0) {
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
    unsigned char st = 1;
    const ssize_t got = read(fds[0], &st, 1);
    if (got == 1) return st; // the run is complete; the child finishes its teardown on its own
    int ws = 0;               // the child ended without reporting: its exit status (or signal) is the program's
    if (waitpid(pid, &ws, 0) < 0) return 1;
    return WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + (WIFSIGNALED(ws) ? WTERMSIG(ws) : 0);
}
