// What the stand-alone labs need from api.hip when they #include a production kernel file directly.
#pragma once
namespace empose {
#ifndef EMPOSE_LAB_HAS_OPTIONS
Options& options() { static Options o; return o; }
#endif
hipError_t allow_dynamic_lds(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
hipError_t coresident_blocks(const void* fn, int threads, size_t lds, int* blocks) {
  if (hipError_t e = allow_dynamic_lds(fn, lds)) return e;
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  (void)hipGetDevice(&dev);
  (void)hipGetDeviceProperties(&prop, dev);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds);
  *blocks = per_cu * prop.multiProcessorCount;
  return e;
}
unsigned* poll_timeout_word() { static unsigned* p = nullptr; if (!p) (void)hipMalloc(&p, 64); return p; }
unsigned poll_timeouts_take() { return 0; }
unsigned poll_timeouts_peek() { return 0; }
}  // namespace empose
