// TEST INFRASTRUCTURE — a stand-in for librccl.so.1 that gathers over a shared-memory file, so that the C ABI's multi-rank tally
// (zk_dist_init / zk_dist_tally with world > 1; include/zkevm_hip.h) runs without N GPUs: RCCL refuses two ranks on one
// device, and this container has none.  Exports the five entry points csrc/dist_tally.hpp binds, with rccl.h's signatures:
//   ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString
// Built twice from this file (tests/fakerccl/build.sh): g++ -> libfakerccl_host.so (buffers are host memory: the CPU backend),
// hipcc -DFAKE_RCCL_HIP -> libfakerccl_hip.so (buffers are device memory, staged with hipMemcpy after synchronising the
// stream: the HIP library, two ranks on one GPU).  Selected through ZK_RCCL_LIB.  Never part of the product.
#include <atomic>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifdef FAKE_RCCL_HIP
#include <hip/hip_runtime.h>
#endif

namespace {
const int MAX_RANKS = 64, SLOT_BYTES = 4096;
struct Shared {
    std::atomic<uint32_t> arrived;     // barrier: ranks that reached the current generation
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> joined;
    uint32_t pad;
    unsigned char slot[MAX_RANKS][SLOT_BYTES];
};
struct Comm {
    Shared* sh;
    int rank, world, fd;
    char path[160];
};
struct UniqueId { char internal[128]; };
const size_t TYPE_BYTES[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};  // ncclInt8, Uint8, Int32, Uint32, Int64, Uint64, Float16, Float32, Float64, Bfloat16

int barrier(Comm* c) {
    const uint32_t gen = c->sh->generation.load(std::memory_order_acquire);
    if (c->sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        c->sh->arrived.store(0, std::memory_order_relaxed);
        c->sh->generation.store(gen + 1, std::memory_order_release);
        return 0;
    }
    const time_t t0 = time(nullptr);
    while (c->sh->generation.load(std::memory_order_acquire) == gen) {
        usleep(50);
        if (time(nullptr) - t0 > 120) return 6;  // a rank never arrived: "remote error"
    }
    return 0;
}
}  // namespace

// What kind of buffers this library's ncclAllGather takes (csrc/dist_tally.hpp reads it: a real RCCL has no such symbol and takes device
// memory): 0 = host pointers (the CPU backend's ranks), 1 = device pointers (the HIP library's).  A mismatch is refused at zk_dist_init.
extern "C" int zk_collective_buffers(void) {
#ifdef FAKE_RCCL_HIP
    return 1;
#else
    return 0;
#endif
}
extern "C" int ncclGetUniqueId(UniqueId* id) {
    memset(id->internal, 0, sizeof id->internal);
    unsigned char rnd[12] = {0};
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, rnd, sizeof rnd) != (ssize_t)sizeof rnd) rnd[0] = (unsigned char)getpid(); close(fd); }
    char* p = id->internal;
    p += sprintf(p, "zkfake-%d-", (int)getpid());
    for (unsigned char b : rnd) p += sprintf(p, "%02x", b);
    return 0;
}
extern "C" int ncclCommInitRank(void** comm, int world, UniqueId id, int rank) {
    if (!comm || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world || strncmp(id.internal, "zkfake-", 7)) return 4;  // invalid argument
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    snprintf(c->path, sizeof c->path, "/tmp/%.100s", id.internal);
    c->fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (c->fd < 0 || ftruncate(c->fd, sizeof(Shared)) != 0) { delete c; return 2; }  // a fresh file reads as zeros: counters start at 0
    c->sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->sh == MAP_FAILED) { close(c->fd); delete c; return 2; }
    c->sh->joined.fetch_add(1);
    if (int r = barrier(c)) return r;  // collective, like the real one
    *comm = c;
    return 0;
}
extern "C" int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
    Comm* c = (Comm*)comm;
    if (!c || dtype < 0 || dtype > 9) return 4;
    const size_t bytes = count * TYPE_BYTES[dtype];
    if (bytes > (size_t)SLOT_BYTES) return 4;
#ifdef FAKE_RCCL_HIP
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (hipMemcpy(c->sh->slot[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
#else
    (void)stream;
    memcpy(c->sh->slot[c->rank], send, bytes);
#endif
    if (int r = barrier(c)) return r;
    for (int k = 0; k < c->world; k++) {
#ifdef FAKE_RCCL_HIP
        if (hipMemcpy((char*)recv + (size_t)k * bytes, c->sh->slot[k], bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
#else
        memcpy((char*)recv + (size_t)k * bytes, c->sh->slot[k], bytes);
#endif
    }
    return barrier(c);  // nobody overwrites its slot before every rank has read it
}
extern "C" int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    if (c->sh->joined.fetch_sub(1) == 1) unlink(c->path);  // the last rank out removes the file
    munmap(c->sh, sizeof(Shared));
    close(c->fd);
    delete c;
    return 0;
}
extern "C" const char* ncclGetErrorString(int r) {
    switch (r) {
    case 0: return "no error";
    case 1: return "unhandled device error (stand-in)";
    case 2: return "unhandled system error (stand-in)";
    case 4: return "invalid argument (stand-in)";
    case 6: return "remote error: a rank never reached the barrier (stand-in)";
    default: return "unknown (stand-in)";
    }
}
