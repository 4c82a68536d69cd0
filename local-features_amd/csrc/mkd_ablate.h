// Timing-only and A/B builds of the describe kernel (mkd_describe.hip), in one place.
//
// A PRODUCT build defines none of the macros below: every switch is then `false`, every hook a no-op that folds away, and
// the kernel source itself carries no preprocessor conditionals -- it reads as the product.  An instrument build
// (tools/ab_build.sh NAME "-DLF_ABLATE_MMA", tools/phase_timing.py, ...) flips one of them; such a build produces WRONG
// descriptors by design and exists only to be timed (NOTEBOOK.md sections 4, 9, 10 quote what they measured).
//
//   LF_ABLATE_BLOAD        no LUT fragment reads from LDS            LF_ABLATE_SPLIT     no f16 hi / lo split of the streams
//   LF_ABLATE_MMA          no matrix instructions (operands kept)    LF_ABLATE_FRONT     no blur, no gradient direction
//   LF_ABLATE_SYNC         no row barrier, no wait for the LDS-DMA   LF_ABLATE_EPILOGUE  no normalisation / whitening
//   LF_ABLATE_FORCE_W4     the 4-wave form at every request size
//   LF_KP_ABLATE_PRODUCER  keypoint mode: the describe waves alone   LF_KP_ABLATE_TAPS   ... everything but a sample's loads
//   LF_KP_PRODUCER_PRIO=n / LF_KP_CONSUMER_PRIO=n                    wave priorities of the two kinds of wave
//   LF_ABLATE_ROWS=n       the row loop walks n of a patch's 32 rows (producers too): the bound of a row-split form
//   LF_PHASE_TIMING        per-wave clocks of the phases of a patch row, left by workgroup 0 in out[wave * 128 + phase]
#pragma once
#include <hip/hip_runtime.h>

namespace lfmkd {
namespace ablate {

#ifdef LF_ABLATE_BLOAD
constexpr bool kNoFragmentReads = true;
#else
constexpr bool kNoFragmentReads = false;
#endif
#ifdef LF_ABLATE_SPLIT
constexpr bool kNoSplit = true;
#else
constexpr bool kNoSplit = false;
#endif
#ifdef LF_ABLATE_MMA
constexpr bool kNoMma = true;
#else
constexpr bool kNoMma = false;
#endif
#ifdef LF_ABLATE_FRONT
constexpr bool kNoFrontEnd = true;
#else
constexpr bool kNoFrontEnd = false;
#endif
#ifdef LF_ABLATE_MUL_FIRST_TAP   // the horizontal blur's first tap as a plain product (round 4's form: a blurred value can be -0.0)
constexpr bool kMulFirstTap = true;
#else
constexpr bool kMulFirstTap = false;
#endif
#ifdef LF_ABLATE_SYNC
constexpr bool kNoRowSync = true;
#else
constexpr bool kNoRowSync = false;
#endif
#ifdef LF_ABLATE_EPILOGUE
constexpr bool kNoEpilogue = true;
#else
constexpr bool kNoEpilogue = false;
#endif
#ifdef LF_ABLATE_FORCE_W4
constexpr bool kForceFourWaves = true;
#else
constexpr bool kForceFourWaves = false;
#endif
#ifdef LF_KP_ABLATE_PRODUCER
constexpr bool kNoProducer = true;
#else
constexpr bool kNoProducer = false;
#endif
#ifdef LF_KP_ABLATE_TAPS
constexpr bool kNoTaps = true;
#else
constexpr bool kNoTaps = false;
#endif
#ifdef LF_KP_PRODUCER_PRIO
constexpr int kProducerPrio = LF_KP_PRODUCER_PRIO;
#else
constexpr int kProducerPrio = 2;      // the product's setting (same-box A/B: +4 % over 0; NOTEBOOK.md 4f)
#endif
#ifdef LF_KP_CONSUMER_PRIO
constexpr int kConsumerPrio = LF_KP_CONSUMER_PRIO;
#else
constexpr int kConsumerPrio = -1;     // the describe waves keep the default priority
#endif

#ifdef LF_ABLATE_ROWS
constexpr int kRows = LF_ABLATE_ROWS;
#else
constexpr int kRows = 32;             // rows of a patch (lib.rs:15 PATCH_SIZE)
#endif

// ---- phase clocks ------------------------------------------------------------------------------------------------
#ifdef LF_PHASE_TIMING
constexpr bool kPhaseTiming = true;
#else
constexpr bool kPhaseTiming = false;
#endif
struct PhaseClock {
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prev = 0;
    __device__ __forceinline__ void start() {
        if constexpr (kPhaseTiming) prev = __builtin_readcyclecounter();
    }
    // time since the previous mark goes to phase i
    __device__ __forceinline__ void mark(int i) {
        if constexpr (kPhaseTiming) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_readcyclecounter();
            clk[i] += t - prev;
            prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // workgroup 0 leaves its waves' clocks where the descriptors would go (every wave of the workgroup calls)
    __device__ __forceinline__ void dump(float *out, int wave, int lane) const {
        if constexpr (kPhaseTiming) {
            __syncthreads();
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 8; ++i) out[wave * 128 + i] = (float)clk[i];
        }
    }
};

}  // namespace ablate
}  // namespace lfmkd
