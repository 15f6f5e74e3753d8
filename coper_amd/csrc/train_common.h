// Shared by the training kernels (coper_train.hip): the dropout keep function.  tests/ restate it in NumPy
// (oracle/coper_train_oracle.py: dropout_keep) so the oracle can be fed the same masks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace coper {

// keep (true) with probability 1 - rate: 24-bit uniform from a counter hash of (seed, step, stage, element)
__host__ __device__ __forceinline__ bool dropout_keep_u32(uint32_t seed, uint32_t step, uint32_t stage, uint32_t idx,
                                                          uint32_t threshold24) {
  uint32_t x = idx * 0x9E3779B1u + seed * 0x85EBCA77u + step * 0xC2B2AE3Du + stage * 0x27D4EB2Fu;
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  x *= 0x297A2D39u;
  x ^= x >> 15;
  return (x >> 8) >= threshold24;
}

__host__ __forceinline__ uint32_t dropout_threshold24(float rate) { return rate <= 0.f ? 0u : (uint32_t)(rate * 16777216.0f); }

}  // namespace coper
