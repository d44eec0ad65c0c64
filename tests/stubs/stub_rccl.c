/* Test double for RCCL (tests/test_exchange_stub.py): exports the entry points librtmi.so looks up at run time and
 * records what was called.  STUB_RCCL_FAIL_AT=n makes the n-th send/recv (1-based) return ncclInternalError. */
#include <stdlib.h>
#include <string.h>
typedef int ncclResult_t;
typedef struct ncclComm *ncclComm_t;
static int g_calls[8];   /* 0 group_start, 1 group_end, 2 send, 3 recv, 4 reduce, 5 comm_count */
static int g_open;       /* groups currently open */
static int g_last_peer = -1;
static size_t g_last_count;
static int p2p;
static int fail_at(void) { const char *e = getenv("STUB_RCCL_FAIL_AT"); return e ? atoi(e) : 0; }
ncclResult_t ncclGroupStart(void) { g_calls[0]++; g_open++; return 0; }
ncclResult_t ncclGroupEnd(void) { g_calls[1]++; g_open--; return 0; }
ncclResult_t ncclSend(const void *b, size_t n, int dt, int peer, ncclComm_t c, void *s) {
  (void)b; (void)dt; (void)c; (void)s;
  g_calls[2]++; g_last_peer = peer; g_last_count = n;
  return ++p2p == fail_at() ? 3 : 0;
}
ncclResult_t ncclRecv(void *b, size_t n, int dt, int peer, ncclComm_t c, void *s) {
  (void)b; (void)dt; (void)c; (void)s;
  g_calls[3]++; g_last_peer = peer; g_last_count = n;
  return ++p2p == fail_at() ? 3 : 0;
}
ncclResult_t ncclReduce(const void *a, void *b, size_t n, int dt, int op, int root, ncclComm_t c, void *s) {
  (void)a; (void)b; (void)dt; (void)op; (void)c; (void)s;
  g_calls[4]++; g_last_peer = root; g_last_count = n;
  return 0;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { (void)c; g_calls[5]++; *n = 4; return 0; }
const char *ncclGetErrorString(ncclResult_t e) { return e == 3 ? "stub internal error" : "stub error"; }
/* inspection */
int stub_calls(int i) { return g_calls[i]; }
int stub_open_groups(void) { return g_open; }
int stub_last_peer(void) { return g_last_peer; }
size_t stub_last_count(void) { return g_last_count; }
