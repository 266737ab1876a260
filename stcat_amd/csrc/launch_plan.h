// Launch plans: the launch sequence of a composite node of the hot path (backbone forward / backward, the
// spatial-temporal encoder, the two decoders, the heads — models/pipeline.py:52-121 in the reference's call order)
// recorded ONCE per input shape as a flat list of C-ABI calls and replayed by stcat_plan_run() in one host call.
// Round 2 issued every one of the ~1700 launches of a step as its own Python -> ctypes call (host floor 29 ms per
// step); a replay costs one hipLaunchKernel per op.
//
// A recorded call is (entry point, argument words).  Every argument of the launch entry points is a device pointer,
// an int / long, a float or the stream, so one 64-bit word per argument holds it and a thunk generated from the entry
// point's own C type unpacks the words again (no libffi, no per-function code).  Pointers into tensors the caller
// passes in per step ("externals": the node's inputs, parameters, upstream gradients) are patched through a
// relocation list (word, external index, byte offset) at the top of every replay; every other pointer was allocated
// from the plan's private memory pool during recording and stays valid for the life of the plan.  Streams are slots
// (0 = the caller's current stream, 1.. = the side streams of the recording); WAIT ops carry the event that orders
// one slot behind another (the weight-gradient stream of the backbone's backward).  YIELD ops hand control back to the
// host in the middle of a replay (the data-parallel reducer's early bucket delivery, stcat_amd/dist.py).
#pragma once

#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace stcat_plan {

typedef int (*ThunkFn)(const uint64_t*);

struct FnEntry {
  const char* name;
  ThunkFn call;
  int nargs;
};

template <class T>
inline T arg_of(uint64_t w) {
  if constexpr (std::is_pointer<T>::value) {
    return reinterpret_cast<T>(static_cast<uintptr_t>(w));
  } else if constexpr (std::is_floating_point<T>::value) {
    float f;
    const uint32_t u = (uint32_t)w;
    memcpy(&f, &u, 4);
    return (T)f;
  } else {
    return (T)(int64_t)w;
  }
}

template <class F, F f>
struct Thunk;
template <class... A, int (*f)(A...)>
struct Thunk<int (*)(A...), f> {
  template <size_t... I>
  static int go(const uint64_t* w, std::index_sequence<I...>) {
    return f(arg_of<A>(w[I])...);
  }
  static int call(const uint64_t* w) { return go(w, std::index_sequence_for<A...>{}); }
  static constexpr int nargs = (int)sizeof...(A);
};

enum OpKind : uint8_t { OP_CALL = 0, OP_WAIT = 1, OP_MEMSET = 2, OP_YIELD = 3 };

struct Op {
  uint8_t kind;
  uint8_t slot;    // CALL / MEMSET: stream slot the op is issued on; WAIT: the slot that waits
  uint8_t slot2;   // WAIT: the slot whose queued work is waited for
  uint8_t pad;
  int32_t fn;      // CALL: index into the entry-point table; YIELD: tag handed back to the host
  uint32_t arg0;   // CALL: first argument word; MEMSET: word of the pointer (bytes in the next word); WAIT: event index
  int32_t nargs;
  int32_t stream_arg;  // CALL: argument position that receives the slot's stream (-1: none)
};

struct Reloc {
  uint32_t word;
  uint32_t ext;
  uint64_t off;
};

struct Plan {
  std::vector<Op> ops;
  std::vector<uint64_t> words;
  std::vector<Reloc> relocs;
  std::vector<void*> events;  // hipEvent_t, created on first replay
  int n_ext = 0;
  int n_slots = 1;
  long replays = 0;
};

}  // namespace stcat_plan
