/* Diagnostics of libctagan_hip.so -- NOT part of the product C ABI of include/ctagan_hip.h.
 *
 * The entry points below ship in the same shared library because the standing GPU checks of this tree call them
 * (tests/test_kernels_gpu.py::test_lds_canary_beside_a_training_step), but no reference call site corresponds to them, a
 * drop-in integration never binds them, and they carry no compatibility promise (ctg_abi_version() does not cover them). */
#ifndef CTAGAN_HIP_DIAG_H
#define CTAGAN_HIP_DIAG_H
#ifdef __cplusplus
extern "C" {
#endif

/* `blocks` workgroups (256 threads, 8 KB of LDS each) fill their LDS with a pattern and re-read it `spins` times (~1 us
 * apart); report (4 + 4 report_cap unsigned words, zeroed by the caller) receives report[0] = the number of words something ELSE
 * changed, then (word index, value found, spin, workgroup) per event.  Run on a second stream beside other launches it detects
 * kernels that write LDS outside their own allocation (tests/test_lds_canary_gpu.py). */
int ctg_lds_canary(int blocks, int spins, unsigned* report, int report_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
