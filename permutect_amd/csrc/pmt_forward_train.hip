// The training forward of the production shape, pmt_forward_kernel<true, ShapeP0X> (and its layered twin), in its own translation unit so that it can be
// compiled with an OPAQUE thread id (pmt_device.hpp: pmt_tid()).  At 128 registers per lane (four waves per SIMD) the compiler
// otherwise hoists every per-lane address derived from the thread id out of the block loop and spills it; a spill reload is a
// vector-memory load, and vmcnt retires in order: each reload then waits for the stash stores issued before it.  With the lane
// coordinates re-derived inside the loop (a few integer instructions) the loop's scratch loads drop from 24 to 11 and the
// kernel from 1.08 to 0.97 ms.  The filter instance does not store a stash and is 2 % FASTER with the hoisting, so it stays in
// pmt_forward.hip, compiled as before.
#define PMT_OPAQUE_TID 1
#define PMT_FORWARD_TRAIN_TU
#include "pmt_forward.hip"
