/* A stand-in for librccl.so whose communicator bring-up NEVER RETURNS (tests/test_comm_gpu.py): what clair_comm_create_timed's
 * deadline is for.  Loaded through CLAIR_AMD_RCCL_LIBRARY; built by the test with gcc.  Only the entry points clair_amd/csrc/comm.hip
 * binds, with RCCL's calling convention (ncclUniqueId is a 128-byte struct passed by value).
 *   CLAIR_FAKE_RCCL_HANG = "all" | "<rank>" : ncclCommInitRank sleeps for ever on those ranks; elsewhere it succeeds at once
 *   CLAIR_FAKE_RCCL_LATE = seconds          : instead of for ever, return after that long (the abandoned helper thread then aborts) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;

int ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return 0; }

int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    const char *hang = getenv("CLAIR_FAKE_RCCL_HANG"), *late = getenv("CLAIR_FAKE_RCCL_LATE");
    (void)nranks; (void)id;
    if (hang && (!strcmp(hang, "all") || atoi(hang) == rank)) {
        if (late) sleep((unsigned)atoi(late));
        else for (;;) sleep(1);
    }
    *comm = malloc(16);
    return 0;
}

int ncclCommDestroy(ncclComm_t comm) { free(comm); return 0; }
int ncclCommAbort(ncclComm_t comm) {
    const char *mark = getenv("CLAIR_FAKE_RCCL_ABORT_MARK");
    if (mark) { FILE *f = fopen(mark, "a"); if (f) { fputs("aborted\n", f); fclose(f); } }
    free(comm);
    return 0;
}
const char *ncclGetErrorString(int result) { (void)result; return "stand-in RCCL error"; }
/* one-rank semantics, in place (comm.hip stages host buffers through one device buffer): nothing to move */
int ncclBroadcast(const void *s, void *r, size_t n, int t, int root, ncclComm_t c, void *st) { (void)s; (void)r; (void)n; (void)t; (void)root; (void)c; (void)st; return 0; }
int ncclAllReduce(const void *s, void *r, size_t n, int t, int op, ncclComm_t c, void *st) { (void)s; (void)r; (void)n; (void)t; (void)op; (void)c; (void)st; return 0; }
int ncclAllGather(const void *s, void *r, size_t n, int t, ncclComm_t c, void *st) { (void)s; (void)r; (void)n; (void)t; (void)c; (void)st; return 0; }
