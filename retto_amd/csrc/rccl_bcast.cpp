// One-time weight broadcast over RCCL through the C ABI ("weights RCCL-broadcast over xGMI once, no
// per-step collectives").  The reference has no distributed code (retto-cli/src/main.rs:80-86 is a serial loop); a host that
// shards pages over the GPUs of a node (one process per GPU) calls rt_rccl_unique_id on rank 0, hands the 128 bytes to the
// other ranks by whatever means it has (a file, a pipe, MPI), and every rank calls rt_broadcast_blobs.  librccl.so is
// loaded lazily: a single-GPU host never touches it.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/retto_hip.h"
#include "common.h"

struct Id128 { char b[128]; };  // ncclUniqueId (passed by value)
namespace {
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // an RCCL the process already holds (e.g. the copy torch loaded) is reused instead of loading a second one beside it
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if (!r.h) r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!r.h) r.h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.h) r.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (r.h) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
      r.Broadcast = (decltype(r.Broadcast))dlsym(r.h, "ncclBroadcast");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.CommDestroy) { dlclose(r.h); r.h = nullptr; }
    }
  }
  return r.h ? &r : nullptr;
}
void set_err(char* err, size_t cap, const std::string& m) { if (err && cap) { strncpy(err, m.c_str(), cap - 1); err[cap - 1] = 0; } }
}  // namespace

extern "C" {

int rt_rccl_unique_id(void* id, size_t cap, char* err, size_t err_cap) {
  if (!id || cap < 128) { set_err(err, err_cap, "rt_rccl_unique_id: need a 128-byte buffer"); return RT_ERR_INVALID; }
  Rccl* r = rccl();
  if (!r) { set_err(err, err_cap, "librccl.so could not be loaded"); return RT_ERR_BACKEND; }
  int rc = r->GetUniqueId(id);
  if (rc != 0) { set_err(err, err_cap, std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error")); return RT_ERR_BACKEND; }
  return RT_OK;
}

int rt_broadcast_blobs(const void* id, int rank, int world, int device_id, int root, int n_blobs, void** data, size_t* lens, char* err,
                       size_t err_cap) {
  if (!id || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || n_blobs < 0 || (n_blobs && (!data || !lens))) {
    set_err(err, err_cap, "rt_broadcast_blobs: bad argument"); return RT_ERR_INVALID;
  }
  Rccl* r = rccl();
  if (!r) { set_err(err, err_cap, "librccl.so could not be loaded"); return RT_ERR_BACKEND; }
  // No timeout: like every RCCL collective this call blocks until all `world` ranks have entered it.  A rank that fails before
  // its ncclBroadcast (hipMalloc, a bad device) leaves the others waiting -- hosts run it under their own watchdog and restart
  // the job (bench.py's launcher kills all ranks after RT_BENCH_RANK_TIMEOUT; examples/retto_dir.cpp exits non-zero when a
  // child does).
  void* comm = nullptr;
  hipStream_t st = nullptr;
  void* dbuf = nullptr;
  std::vector<void*> mine;  // buffers this call allocated (released on failure)
  int prev_dev = -1;
  (void)hipGetDevice(&prev_dev);   // the caller's current device is restored on every exit path
  auto fail = [&](const std::string& m) {
    set_err(err, err_cap, m);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    for (void* p : mine) free(p);
    if (rank != root) for (int i = 0; i < n_blobs; i++) data[i] = nullptr;
    if (dbuf) (void)hipFree(dbuf);
    if (st) (void)hipStreamDestroy(st);
    if (comm) r->CommDestroy(comm);
    return RT_ERR_BACKEND;
  };
  if (hipSetDevice(device_id) != hipSuccess) return fail("hipSetDevice failed");
  Id128 uid; memcpy(uid.b, id, 128);
  int rc = r->CommInitRank(&comm, world, uid, rank);
  if (rc != 0) { comm = nullptr; return fail(std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error")); }
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return fail("hipStreamCreate failed"); }
  // sizes first (8 bytes per blob), then every blob through one device staging buffer
  std::vector<unsigned long long> sz((size_t)std::max(n_blobs, 1), 0);
  if (rank == root) for (int i = 0; i < n_blobs; i++) sz[(size_t)i] = lens[i];
  size_t cap = std::max<size_t>(sz.size() * 8, 256);
  if (hipMalloc(&dbuf, cap) != hipSuccess) { dbuf = nullptr; return fail("hipMalloc failed"); }
  auto bcast = [&](void* host, size_t bytes) -> bool {
    if (bytes == 0) return true;
    if (bytes > cap) {
      (void)hipFree(dbuf); dbuf = nullptr;
      if (hipMalloc(&dbuf, bytes) != hipSuccess) { dbuf = nullptr; return false; }
      cap = bytes;
    }
    if (rank == root && hipMemcpyAsync(dbuf, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    if (r->Broadcast(dbuf, dbuf, bytes, /* ncclUint8 */ 1, root, comm, st) != 0) return false;
    if (rank != root && hipMemcpyAsync(host, dbuf, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    return hipStreamSynchronize(st) == hipSuccess;
  };
  if (!bcast(sz.data(), (size_t)n_blobs * 8)) return fail("broadcast of the blob sizes failed");
  for (int i = 0; i < n_blobs; i++) {
    if (rank != root) {
      lens[i] = (size_t)sz[(size_t)i];
      data[i] = malloc(std::max<size_t>(lens[i], 1));
      if (!data[i]) return fail("out of host memory");
      mine.push_back(data[i]);
    }
    if (!bcast(data[i], (size_t)sz[(size_t)i])) return fail("broadcast of blob " + std::to_string(i) + " failed");
  }
  (void)hipFree(dbuf);
  (void)hipStreamDestroy(st);
  r->CommDestroy(comm);
  if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
  return RT_OK;
}

}  // extern "C"
