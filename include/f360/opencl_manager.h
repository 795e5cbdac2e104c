// f360/opencl_manager.h -- drop-in for the reference's src/opencl_manager.h.
//
// The reference's transform classes borrow an `OpenCLManager*` and its callers
// use a handful of OpenCL C++ names directly:
//   cl::Buffer(context, flags, bytes), buffer(), cl::copy(queue, begin, end, buffer),
//   cl::copy(queue, buffer, begin, end), clFlush/clFinish(queue()),
//   command_queue.finish(), OpenCLManager::GetCLErrorString, CL_SUCCESS,
//   CL_MEM_READ_WRITE / CL_MEM_READ_ONLY
// (src/video_server.cc:224-232,298-303,342-345; src/run_satlogrectilinear.cc
// :381-412).  This header provides exactly those names on top of the C ABI of
// f360.h, so such call sequences compile unchanged against the HIP engine:
// `cl_mem` becomes a plain device pointer and the "command queue" is the
// context's in-order HIP stream.
#pragma once

#include <cstddef>
#include <cstdint>
#include <iostream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../f360.h"

typedef void *cl_mem;  // device pointer
typedef int cl_int;
typedef unsigned long cl_mem_flags;

#ifndef CL_SUCCESS
#define CL_SUCCESS 0
#define CL_MEM_READ_WRITE (1 << 0)
#define CL_MEM_WRITE_ONLY (1 << 1)
#define CL_MEM_READ_ONLY (1 << 2)
#endif

namespace cl {

// The engine context (device + in-order stream); shared by copies of the handle.
class Context {
 public:
  Context() = default;
  explicit Context(f360_ctx *raw) : ctx_(raw, [](f360_ctx *c) { f360_ctx_destroy(c); }) {}
  f360_ctx *get() const { return ctx_.get(); }
  f360_ctx *operator()() const { return ctx_.get(); }

 private:
  std::shared_ptr<f360_ctx> ctx_;
};

class CommandQueue {
 public:
  CommandQueue() = default;
  explicit CommandQueue(const Context &c) : context_(c) {}
  cl_int finish() const { return context_.get() ? f360_sync(context_.get()) : F360_ERR_NOT_INITIALIZED; }
  cl_int flush() const { return CL_SUCCESS; }  // the stream submits eagerly
  const CommandQueue &operator()() const { return *this; }
  f360_ctx *ctx() const { return context_.get(); }

 private:
  Context context_;
};

// cl::Buffer(context, flags, bytes): RAII device allocation (reference-counted
// like the OpenCL wrapper, so copies are cheap and the last one frees).
class Buffer {
 public:
  Buffer() = default;
  Buffer(const Context &context, cl_mem_flags /*flags*/, std::size_t bytes, void * = nullptr,
         cl_int *err = nullptr) {
    void *p = nullptr;
    const int st = context.get() ? f360_malloc(context.get(), bytes, &p) : F360_ERR_NOT_INITIALIZED;
    if (err) *err = st;
    if (st == F360_OK) {
      Context keep = context;
      mem_ = std::shared_ptr<void>(p, [keep](void *q) { f360_free(keep.get(), q); });
      bytes_ = bytes;
    }
  }
  cl_mem operator()() const { return mem_.get(); }
  std::size_t size() const { return bytes_; }

 private:
  std::shared_ptr<void> mem_;
  std::size_t bytes_ = 0;
};

// host -> device, blocking (cl::copy(queue, startIterator, endIterator, buffer))
template <class It>
inline cl_int copy(const CommandQueue &queue, It begin, It end, Buffer &buffer) {
  using T = typename std::iterator_traits<It>::value_type;
  const std::size_t n = static_cast<std::size_t>(std::distance(begin, end));
  if (n == 0) return CL_SUCCESS;
  return f360_memcpy_h2d(queue.ctx(), buffer(), &*begin, n * sizeof(T));
}
// device -> host, blocking (cl::copy(queue, buffer, startIterator, endIterator))
template <class It>
inline cl_int copy(const CommandQueue &queue, const Buffer &buffer, It begin, It end) {
  using T = typename std::iterator_traits<It>::value_type;
  const std::size_t n = static_cast<std::size_t>(std::distance(begin, end));
  if (n == 0) return CL_SUCCESS;
  return f360_memcpy_d2h(queue.ctx(), &*begin, buffer(), n * sizeof(T));
}

}  // namespace cl

inline cl_int clFlush(const cl::CommandQueue &q) { return q.flush(); }
inline cl_int clFinish(const cl::CommandQueue &q) { return q.finish(); }

// src/opencl_manager.h:8-22.  The GL-sharing members of the reference
// (gl_context / gl_display, used only by the SDL client) have no counterpart.
class OpenCLManager {
 public:
  cl::Context context;
  cl::CommandQueue command_queue;
  int device_index = 0;

  OpenCLManager() = default;
  ~OpenCLManager() = default;

  // src/opencl_manager.cc:7-67: devices[0], one in-order queue.  Returns 0 on
  // success like the reference.
  int InitializeContext() {
    f360_ctx *raw = nullptr;
    const int st = f360_ctx_create(device_index, &raw);
    if (st != F360_OK) {
      std::cerr << "Failed to create device context: " << f360_last_error_string() << std::endl;
      return st;
    }
    context = cl::Context(raw);
    command_queue = cl::CommandQueue(context);
    return 0;
  }

  static std::string GetCLErrorString(cl_int error) {
    return std::string(f360_status_string(error));
  }
};
