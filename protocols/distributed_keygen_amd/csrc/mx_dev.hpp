// Developer build switches of the kernels — ALL of them, in one place.  The shipped library is built with none of them
// (build.py's FLAGS carry no -DMX_DEV_*; tests/test_instances.py checks that, and that no other MX_DEV_ name exists in
// these sources).  They are set through tools/build_variant.py <name> -DMX_DEV_...  which writes
// build/variants/<name>.so, loaded through the MX_LIBRARY environment variable of _lib.py.  Nothing else in the sources
// may branch on a macro.
//
//   switch                          effect                                                       read by
//   MX_DEV_TS_COMPILER_RELEASE      the time-sliced pair kernel publishes a group's next unit    tools/prove_handover_guard.sh,
//                                   with the compiler's release store instead of the explicit    tests/test_gpu_handover.py (which
//                                   s_waitcnt / buffer_wbl2 / s_waitcnt — the sequence that      must FAIL on such a build)
//                                   lost groups in rounds 3-4 (mx_powmod_n2_split.hpp)
//   MX_DEV_TS_NO_A_FENCE            wavefront A of a pair does not release its own stores at     tools/ts_handover_check.py
//                                   agent scope before B pushes the unit
//   MX_DEV_TS_MIN_WAVES=n           occupancy bound (waves per SIMD) of the 9- and 3-limb        tools/ts_probe.py
//                                   time-sliced instances, default 3
//   MX_DEV_TS_TRACE                 every unit of a time-sliced launch records when, on which    tools/ts_trace.py
//                                   pair and on which XCD it ran (four words behind the queues)
//   MX_DEV_BI_TRACE                 shader-clock cycles per phase of the bipartite form's        tools/bi_phase_probe.py
//                                   products, pair 0 of workgroup 0 (mx_bimont.hpp)
//   MX_DEV_BP_TRACE                 shader-clock cycles per phase and role of the five-wavefront pair    tools/bp_phase_probe.py
//                                   kernel, workgroup 0 (mx_bipair.hpp)
//   MX_DEV_PRIVATE_PAD_WORDS=n      every lane of the one-wavefront pair kernel keeps n tagged   tools/concurrency_census.py
//                                   words in a private segment and checks them at the end
//   MX_DEV_LDS_PAD_WORDS=n          n more words between the LDS scratch of a wavefront's        tools/lds_stride_ab.sh
//                                   groups, default 0 (mx_mont.hpp)
//   MX_DEV_AUX_WAVE_PRIO=p          s_setprio level of the short kernels, default 3, 0 = none    tools/gpu_session.sh prio_ab
//                                   (mx_prio.hpp)
#pragma once

#ifndef MX_DEV_TS_MIN_WAVES
#define MX_DEV_TS_MIN_WAVES 3
#endif
#ifndef MX_DEV_LDS_PAD_WORDS
#define MX_DEV_LDS_PAD_WORDS 0
#endif
#ifndef MX_DEV_AUX_WAVE_PRIO
#define MX_DEV_AUX_WAVE_PRIO 3
#endif

#if defined(MX_DEV_TS_COMPILER_RELEASE) || defined(MX_DEV_TS_NO_A_FENCE) || defined(MX_DEV_TS_TRACE) || defined(MX_DEV_BI_TRACE) || defined(MX_DEV_BP_TRACE) || \
    defined(MX_DEV_PRIVATE_PAD_WORDS) || MX_DEV_TS_MIN_WAVES != 3 || MX_DEV_LDS_PAD_WORDS != 0 || MX_DEV_AUX_WAVE_PRIO != 3
#define MX_DEV_BUILD 1          // a developer variant: never the library a release is built from
#else
#define MX_DEV_BUILD 0
#endif
