// multi_job.h -- many small launches of one kernel as ONE launch.
//
// The finishers of a decode call are the same few kernels once per column and stripe (a lineitem call: 12 x the seven kernels of
// the decimal chain, 15 x the UTF-8 checks, ...), each far too small to fill the machine: launched one by one they cost their
// launch latency (4-5 us) each, back to back.  Here the arguments of every such launch are filed as a job; jobs of the same kernel
// and the same pipeline stage leave together as one grid (blockIdx.y = job, blockIdx.x = the block of that job's own grid).
#pragma once
#include <type_traits>
#include <utility>

#define MJ_SLOTS 14
struct MJob {
  uint64_t a[MJ_SLOTS];  // the arguments, 8 bytes each
  uint32_t nblocks;      // the job's own grid
  uint32_t pad;
};

template <typename T>
__device__ __forceinline__ T mj_get(uint64_t v) {
  static_assert(sizeof(T) <= 8, "multi_job: arguments are at most 8 bytes");
  T t;
  __builtin_memcpy(&t, &v, sizeof(T));
  if constexpr (std::is_pointer<T>::value) {
    // every pointer a job carries is device global memory (see as_global)
    return t ? (T)as_global((void*)const_cast<typename std::remove_const<typename std::remove_pointer<T>::type>::type*>(t)) : t;
  } else {
    return t;
  }
}
template <typename T>
inline uint64_t mj_put(T t) {
  static_assert(sizeof(T) <= 8, "multi_job: arguments are at most 8 bytes");
  uint64_t v = 0;
  __builtin_memcpy(&v, &t, sizeof(T));
  return v;
}
template <typename F>
struct MjSig;
template <typename... A>
struct MjSig<void (*)(A...)> {
  static constexpr size_t n = sizeof...(A);
  template <void (*Body)(A...), size_t... I>
  static __device__ __forceinline__ void call(const MJob& j, std::index_sequence<I...>) {
    Body(mj_get<A>(j.a[I])...);
  }
  template <typename... B, size_t... I>
  static void pack(MJob& j, std::index_sequence<I...>, B... b) {
    static_assert(sizeof...(B) == sizeof...(A), "multi_job: wrong number of arguments");
    ((j.a[I] = mj_put<A>((A)b)), ...);
  }
};
template <auto Body, int BS>
__global__ void __launch_bounds__(BS) mj_kernel(const MJob* jobs) {
  const MJob j = jobs[blockIdx.y];
  if (blockIdx.x >= j.nblocks) return;
  using S = MjSig<decltype(Body)>;
  static_assert(S::n <= MJ_SLOTS, "multi_job: too many arguments");
  S::template call<Body>(j, std::make_index_sequence<S::n>{});
}
