/* A FUNCTIONAL stand-in for RCCL that moves data between processes through files (tests/test_gpu_scene_programs.py:
 * the pool's boxes have one GPU and RCCL refuses two ranks on one device, so the N-rank path of DistributedMain /
 * rtmi_gather / rtmi_reduce_sum cannot run over the real library there).  LD_PRELOADed into the scene program, it
 * answers the entry points the program and librtmi.so use: point-to-point messages are files in STUB_RCCL_DIR named
 * after (source, destination, sequence number), written under a temporary name and renamed; a receive polls for its
 * file.  ncclReduce(sum) adds the ranks' buffers on the root in rank order.  Everything is synchronous: the stream is
 * drained before a buffer is read and the data is in place when the call returns. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm { int rank, n; unsigned seq_to[64], seq_from[64]; };
typedef struct ncclComm *ncclComm_t;

static const char *dir(void) { const char *d = getenv("STUB_RCCL_DIR"); return d ? d : "/tmp"; }
static size_t bytes_of(int dt, size_t n) { return n * (dt == 7 /* ncclFloat */ ? 4 : dt == 8 /* ncclDouble */ ? 8 : 1); }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof(*id)); return 0; }
/* Like the real one, ncclCommInitRank returns only when every rank has called it (the caller deletes its rendezvous file
 * right afterwards): each rank leaves a join_<rank> file and waits for the others'. */
ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId id, int rank) {
  (void)id;
  if (n > 64) return 4;
  char path[512];
  snprintf(path, sizeof(path), "%s/join_%d", dir(), rank);
  FILE *f = fopen(path, "wb");
  if (!f) return 2;
  fclose(f);
  for (int r = 0; r < n; r++) {
    snprintf(path, sizeof(path), "%s/join_%d", dir(), r);
    int tries = 0;
    while (access(path, F_OK) != 0) {
      if (++tries > 6000) return 3;
      struct timespec ts = {0, 50 * 1000 * 1000};
      nanosleep(&ts, NULL);
    }
  }
  *c = (ncclComm_t)calloc(1, sizeof(**c));
  (*c)->rank = rank, (*c)->n = n;
  return 0;
}
/* (the join files stay: a rank that only sends may be done before a slow rank has looked for its file) */
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->n; return 0; }
ncclResult_t ncclGroupStart(void) { return 0; }
ncclResult_t ncclGroupEnd(void) { return 0; }
const char *ncclGetErrorString(ncclResult_t e) { return e ? "file-RCCL stub error" : "no error"; }

static ncclResult_t put(const void *dbuf, size_t bytes, int from, int to, unsigned seq, hipStream_t s) {
  char tmp[640], path[512];
  snprintf(path, sizeof(path), "%s/msg_%d_%d_%u", dir(), from, to, seq);
  snprintf(tmp, sizeof(tmp), "%s.tmp", path);
  void *h = malloc(bytes ? bytes : 1);
  if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h, dbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  FILE *f = fopen(tmp, "wb");
  if (!f || fwrite(h, 1, bytes, f) != bytes) return 2;
  fclose(f);
  free(h);
  return rename(tmp, path) == 0 ? 0 : 2;
}
static ncclResult_t get(void *h, size_t bytes, int from, int to, unsigned seq) {
  char path[512];
  snprintf(path, sizeof(path), "%s/msg_%d_%d_%u", dir(), from, to, seq);
  for (int tries = 0; tries < 6000; tries++) {  /* up to five minutes */
    FILE *f = fopen(path, "rb");
    if (f) {
      const size_t got = fread(h, 1, bytes, f);
      fclose(f);
      unlink(path);
      return got == bytes ? 0 : 2;
    }
    struct timespec ts = {0, 50 * 1000 * 1000};
    nanosleep(&ts, NULL);
  }
  return 3;
}
ncclResult_t ncclSend(const void *buf, size_t n, int dt, int peer, ncclComm_t c, hipStream_t s) {
  return put(buf, bytes_of(dt, n), c->rank, peer, c->seq_to[peer]++, s);
}
ncclResult_t ncclRecv(void *buf, size_t n, int dt, int peer, ncclComm_t c, hipStream_t s) {
  const size_t bytes = bytes_of(dt, n);
  void *h = malloc(bytes ? bytes : 1);
  ncclResult_t e = get(h, bytes, peer, c->rank, c->seq_from[peer]++);
  if (!e && (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(buf, h, bytes, hipMemcpyHostToDevice) != hipSuccess)) e = 1;
  free(h);
  return e;
}
ncclResult_t ncclReduce(const void *send, void *recv, size_t n, int dt, int op, int root, ncclComm_t c, hipStream_t s) {
  if (dt != 7 || op != 0 /* ncclSum */) return 4;
  if (c->rank != root) return put(send, n * 4, c->rank, root, c->seq_to[root]++, s);
  float *acc = (float *)malloc(n * 4 + 4), *in = (float *)malloc(n * 4 + 4);
  if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(acc, send, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int r = 0; r < c->n; r++) {
    if (r == root) continue;
    ncclResult_t e = get(in, n * 4, r, root, c->seq_from[r]++);
    if (e) return e;
    for (size_t i = 0; i < n; i++) acc[i] += in[i];
  }
  const hipError_t he = hipMemcpy(recv, acc, n * 4, hipMemcpyHostToDevice);
  free(acc), free(in);
  return he == hipSuccess ? 0 : 1;
}
