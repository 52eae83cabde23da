/*
 * tfhe_hip_internal.hpp -- option ids that exist for the repository's own tests and measurements and are NOT part of the public ABI
 * (include/tfhe_hip.h): a packaging build exposes no switch that changes behaviour for test purposes.  They go through
 * tfhe_ctx_set_option / tfhe_ctx_get_option like the public ones; the Python mirror (go-tfhe_amd/_binding.py) knows their numbers.
 */
#ifndef TFHE_HIP_INTERNAL_H
#define TFHE_HIP_INTERNAL_H

enum {
    TFHE_OPT_CLONE_FORCE_HOST = 10,  /* tests: 1 = clones OF this context take the host-staged path (the fallback of devices that are not
                                        peers) whatever the devices are, so that the path is exercised on a one-GPU box                    */
    TFHE_OPT_COMBINE_EXIT_NONE = 15  /* read-only, 15 ... 20: how the leaders' gathering waits ended, counted per launch they led: nobody to
                                        wait for (15), the previous launch was long ago (16), every caller back (17), batch full (18), a
                                        quiet window passed (19), four windows in all (20) -- tools/combine_bench.cpp                       */
};

#endif
