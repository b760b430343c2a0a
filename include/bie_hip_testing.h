/* bie_hip_testing.h -- fault-injection hooks of libbie_hip.so.  TEST INFRASTRUCTURE, not part of the drop-in boundary
 * (include/bie_hip.h): nothing under bitorch_engine/ calls these; tests/test_gpu_parity.py does, to prove that the in-kernel
 * hand-offs fail loudly. */
#ifndef BIE_HIP_TESTING_H
#define BIE_HIP_TESTING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Make the split-K reducers of subsequent launches expect tag ^ tag_skew and give up after spin_limit polls; the dependency
 * waits of list launches give up after spin_limit polls too.  (0, 0) restores normal operation.  Forges a stale granule /
 * a producer that never finishes. */
void bie_test_forge_reducer(unsigned tag_skew, int spin_limit);
/* Make dependent list entries of subsequent launches wait for `extra` more producer tiles than exist (a producer that never
 * finishes); 0 restores normal operation. */
void bie_test_forge_dependency(int extra);

#ifdef __cplusplus
}
#endif
#endif
