// mtrpl_fp32_pass.hpp -- the fp32 build of the team kernel for the precision sweep of
// BASELINE.json configs[4] / SURVEY.md 8d "Config 5".
//
// The device code of the team kernel (solver_device.hpp, solve_kernel.hpp's integrator pieces,
// solver_block.hpp, solve_kernel_blk.hpp) is compiled a SECOND time inside namespace mtrpl32 with
// every `double` read as `float` (state, backward differences, Jacobian blocks, norms, step
// control, time).  What stays fp64 in both passes is memory shared with the host: descriptors,
// sample times and output samples (`f64`).  Floating literals are C++ doubles, so an expression
// that contains one is evaluated in fp64 and rounded to fp32 when stored: every stored
// intermediate is fp32, some individual operations are better than fp32 -- the sweep therefore
// bounds what a pure-fp32 kernel could reach from above.  Throughput of this build is not a
// claim (the promoted literals cost fp64 instructions and conversions).
#ifndef MTRPL_FP32_PASS_HPP
#define MTRPL_FP32_PASS_HPP

#include "solve_kernel_blk.hpp"  // the fp64 pass (and every system header) first

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc++11-narrowing"
#pragma clang diagnostic ignored "-Wimplicit-const-int-float-conversion"
#pragma clang diagnostic ignored "-Wliteral-range"
#define MTRPL_FP32_PASS 1
#define double float
#define double2 float2
namespace mtrpl32 {
#include "solver_device.hpp"
#include "solve_kernel.hpp"
#include "solver_block.hpp"
#include "solve_kernel_blk.hpp"
}  // namespace mtrpl32
#undef double2
#undef double
#undef MTRPL_FP32_PASS
#pragma clang diagnostic pop

#endif
