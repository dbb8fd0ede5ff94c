// In-kernel phase stamps of the diagnostic builds (scripts/build_timing.sh); not part of the product library.
#pragma once
// Diagnostic builds only (-DPESR_TIMING, scripts/kernel_phases.py): thread 0 of every workgroup stamps the 100 MHz real-time
// counter at phase boundaries into a per-kernel device array that a debug entry point copies out.  Never defined in the product build.
#ifdef PESR_TIMING
#define PESR_TIMING_SLOTS 8
#define PESR_STAMP(buf, slot) do { if (threadIdx.x == 0) (buf)[blockIdx.x * PESR_TIMING_SLOTS + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PESR_STAMP_CLK(buf, slot) do { if (threadIdx.x == 0) (buf)[blockIdx.x * PESR_TIMING_SLOTS + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PESR_STAMP(buf, slot) do { } while (0)
#define PESR_STAMP_CLK(buf, slot) do { } while (0)
#endif

