// One-shot peer-to-peer exchange for the head-group-parallel decode step: the step's single collective
// moves 3 KiB (all-gather of a rank's context slice) or 16 KiB (all-reduce of the [hidden] fp32 partials) per rank --
// latency-bound, far below what a ring collective amortises.  Every rank owns one exchange buffer that ALL ranks map
// (hipIpc handles across processes; the same pointer inside one process); a step's exchange is ONE kernel per rank:
//   1. write my slice into slot [parity][my rank] of every rank's buffer (direct stores over xGMI),
//   2. system-scope release, then raise flag [parity][my rank] = epoch in every rank's buffer,
//   3. wait until all flags of MY buffer show this epoch (bounded spin), system-scope acquire,
//   4. copy / sum the n slots of my buffer to the output.
// Slots and flags alternate between two parities: a rank can be at most one exchange ahead of its slowest peer (its
// exchange e+1 completes only after every peer has raised flag e+1, i.e. has finished reading the slots of e), so the
// slot it overwrites in exchange e+2 is no longer being read.  The epoch lives in the buffer and is advanced by the kernel
// itself, so a captured hipGraph replays correctly.  The reference has no collective (it is single-GPU); RCCL through
// torch.distributed stays available as the baseline (palu_amd/kernel/head_parallel.py: DistExchange).
// The buffer is UNCACHED device memory (hipExtMallocWithFlags, hipDeviceMallocUncached; fine-grained as the fallback): a
// running kernel polls flags and then reads slots that PEER GPUs write over xGMI -- ordinary (coarse-grained) hipMalloc
// memory gives no coherence inside a kernel (the owner's L2 may keep serving stale lines), which is why RCCL allocates its
// flags and staging buffers the same way.  A wait that times out poisons the output (all-ones bytes = NaN in fp16 / fp32)
// instead of reducing whatever the slots hold.
#include <string.h>

#include "palu_common.h"

namespace {

constexpr int EX_THREADS = 256;
constexpr int EX_MAX_RANKS = 64;
constexpr size_t EX_CTRL_BYTES = 256;                  // [0] epoch, [1] error (a timed-out wait), rest reserved
constexpr size_t EX_FLAG_BYTES = 2 * EX_MAX_RANKS * 4; // [parity][rank] epochs

constexpr size_t EX_DATA_OFF = EX_CTRL_BYTES + EX_FLAG_BYTES;

struct ExParams {
  const char* src;
  size_t bytes;          // bytes of one rank's slice (multiple of 16)
  char* const* peers;    // device array [n]: base of every rank's exchange buffer (own included)
  int rank, n;
  size_t slot_bytes;
  char* out;             // mode 0: [n][bytes] gathered; mode 1: [bytes] = sum over ranks, fp32
  int mode;
  unsigned max_spin;
};

__global__ __launch_bounds__(EX_THREADS) void exchange_kernel(ExParams p) {
  const int tid = threadIdx.x;
  char* mine = p.peers[p.rank];
  unsigned* ctrl = reinterpret_cast<unsigned*>(mine);
  __shared__ unsigned e_sh, timeout_sh;
  if (tid == 0) {
    timeout_sh = 0u;
    const unsigned e = ctrl[0] + 1;    // only this rank's exchange kernels touch its epoch word, one at a time (stream order)
    ctrl[0] = e;
    e_sh = e;
  }
  __syncthreads();
  const unsigned e = e_sh;
  const int par = (int)(e & 1u);
  const size_t nchunk = p.bytes / 16;
  // 1. my slice into every rank's slot [par][rank]
  for (int q = 0; q < p.n; ++q) {
    u32x4* dst = reinterpret_cast<u32x4*>(p.peers[q] + EX_DATA_OFF + ((size_t)par * p.n + p.rank) * p.slot_bytes);
    const u32x4* s = reinterpret_cast<const u32x4*>(p.src);
    for (size_t i = tid; i < nchunk; i += EX_THREADS) dst[i] = s[i];
  }
  // 2. release at system scope (write back what sits in this XCD's L2), then the flags
  __threadfence_system();
  __syncthreads();
  if (tid < p.n) {
    unsigned* f = reinterpret_cast<unsigned*>(p.peers[tid] + EX_CTRL_BYTES) + par * EX_MAX_RANKS + p.rank;
    __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. every rank's flag in my buffer
  if (tid < p.n) {
    unsigned* f = reinterpret_cast<unsigned*>(mine + EX_CTRL_BYTES) + par * EX_MAX_RANKS + tid;
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > p.max_spin) {          // a peer never arrived: report instead of hanging the GPU
        ctrl[1] = e;
        timeout_sh = 1u;
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
  // 4. the n slots of my buffer -> out (poison after a timed-out wait: the slots of the missing peers hold old data)
  const char* data = mine + EX_DATA_OFF + (size_t)par * p.n * p.slot_bytes;
  if (timeout_sh) {
    const u32x4 bad = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const size_t nout = p.mode == 0 ? nchunk * p.n : nchunk;
    for (size_t i = tid; i < nout; i += EX_THREADS) reinterpret_cast<u32x4*>(p.out)[i] = bad;
  } else if (p.mode == 0) {
    for (int q = 0; q < p.n; ++q) {
      const u32x4* s = reinterpret_cast<const u32x4*>(data + (size_t)q * p.slot_bytes);
      u32x4* d = reinterpret_cast<u32x4*>(p.out + (size_t)q * p.bytes);
      for (size_t i = tid; i < nchunk; i += EX_THREADS) d[i] = s[i];
    }
  } else {
    // fp32 sum in rank order: the same arithmetic on every rank
    for (size_t i = tid; i < nchunk; i += EX_THREADS) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < p.n; ++q) acc += *reinterpret_cast<const f32x4*>(data + (size_t)q * p.slot_bytes + i * 16);
      *reinterpret_cast<f32x4*>(p.out + i * 16) = acc;
    }
  }
}

}  // namespace

extern "C" size_t palu_exchange_bytes(int nranks, size_t slot_bytes) {
  if (nranks <= 0 || nranks > EX_MAX_RANKS || slot_bytes == 0 || slot_bytes % 16) return 0;
  return EX_DATA_OFF + 2 * (size_t)nranks * slot_bytes;
}

// Setup (not part of a step; allocates): zero-initialised device memory that other processes can map.
extern "C" int palu_exchange_alloc(size_t bytes, void** ptr) {
  PALU_REQUIRE(ptr && bytes > 0, PALU_ERR_ARG, "exchange_alloc: bad arguments");
  hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);
  }
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_alloc: hipExtMallocWithFlags(%zu, uncached / fine-grained) failed: %s", bytes,
               hipGetErrorString(e));
  e = hipMemset(*ptr, 0, bytes);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_alloc: hipMemset failed: %s", hipGetErrorString(e));
  e = hipDeviceSynchronize();
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_alloc: %s", hipGetErrorString(e));
  return PALU_OK;
}

extern "C" int palu_exchange_free(void* ptr) {
  if (!ptr) return PALU_OK;
  hipError_t e = hipFree(ptr);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_free: %s", hipGetErrorString(e));
  return PALU_OK;
}

extern "C" size_t palu_exchange_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int palu_exchange_export(void* ptr, void* handle_out_host) {
  PALU_REQUIRE(ptr && handle_out_host, PALU_ERR_ARG, "exchange_export: null pointer");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_export: hipIpcGetMemHandle failed: %s", hipGetErrorString(e));
  memcpy(handle_out_host, &h, sizeof(h));
  return PALU_OK;
}

extern "C" int palu_exchange_import(const void* handle_host, void** ptr_out) {
  PALU_REQUIRE(handle_host && ptr_out, PALU_ERR_ARG, "exchange_import: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_host, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_import: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
  return PALU_OK;
}

extern "C" int palu_exchange_close(void* imported_ptr) {
  if (!imported_ptr) return PALU_OK;
  hipError_t e = hipIpcCloseMemHandle(imported_ptr);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_close: %s", hipGetErrorString(e));
  return PALU_OK;
}

static int exchange_launch(const void* src, size_t bytes, const void* peers_dev, int rank, int nranks, size_t slot_bytes,
                           void* out, int mode, palu_stream_t stream) {
  PALU_REQUIRE(src && peers_dev && out, PALU_ERR_ARG, "exchange: null pointer");
  PALU_REQUIRE(nranks > 0 && nranks <= EX_MAX_RANKS && rank >= 0 && rank < nranks, PALU_ERR_ARG, "exchange: bad rank %d of %d",
               rank, nranks);
  PALU_REQUIRE(bytes > 0 && bytes % 16 == 0 && bytes <= slot_bytes && slot_bytes % 16 == 0, PALU_ERR_ARG,
               "exchange: slice of %zu bytes (multiple of 16) must fit the slot of %zu", bytes, slot_bytes);
  PALU_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0, PALU_ERR_ARG, "exchange: src / out must be 16-byte aligned");
  ExParams p;
  p.src = (const char*)src; p.bytes = bytes; p.peers = (char* const*)peers_dev; p.rank = rank; p.n = nranks;
  p.slot_bytes = slot_bytes; p.out = (char*)out; p.mode = mode;
  p.max_spin = 1u << 20;               // x (s_sleep 8 + a system-scope load) ~ 1 s: a hung peer surfaces as an error word
  hipLaunchKernelGGL(exchange_kernel, dim3(1), dim3(EX_THREADS), 0, (hipStream_t)stream, p);
  PALU_LAUNCH_CHECK();
  return PALU_OK;
}

extern "C" int palu_exchange_allgather(const void* src, size_t bytes, const void* peers_dev, int rank, int nranks,
                                       size_t slot_bytes, void* out, palu_stream_t stream) {
  return exchange_launch(src, bytes, peers_dev, rank, nranks, slot_bytes, out, 0, stream);
}

extern "C" int palu_exchange_allreduce_f32(const void* src, size_t bytes, const void* peers_dev, int rank, int nranks,
                                           size_t slot_bytes, void* out, palu_stream_t stream) {
  return exchange_launch(src, bytes, peers_dev, rank, nranks, slot_bytes, out, 1, stream);
}

// host read of the buffer's control words (after a synchronisation): *epoch = exchanges done, *error = epoch of a wait that
// timed out (0 = none)
extern "C" int palu_exchange_status(const void* buffer, unsigned* epoch_host, unsigned* error_host) {
  PALU_REQUIRE(buffer && epoch_host && error_host, PALU_ERR_ARG, "exchange_status: null pointer");
  unsigned w[2];
  hipError_t e = hipMemcpy(w, buffer, sizeof(w), hipMemcpyDeviceToHost);
  PALU_REQUIRE(e == hipSuccess, PALU_ERR_LAUNCH, "exchange_status: %s", hipGetErrorString(e));
  *epoch_host = w[0];
  *error_host = w[1];
  return PALU_OK;
}
