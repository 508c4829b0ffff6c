// crypto_amd/csrc/dock_g2_stub.cpp — DEVELOPMENT ONLY (`make g1only`): G2 entry points that report "not built" so that
// G1 kernel iterations do not wait for the 5-minute G2 translation unit.  Never part of the shipped libdock_gpu.so
// (`make` / __graft_entry__.build() link dock_g2.o).
#include "../../include/dock_gpu.h"
extern "C" {
int32_t dgpu_fold_g2(const uint64_t *, size_t, uint64_t *) { return DGPU_E_NODEVICE; }
int32_t dgpu_msm_g2(const uint64_t *, const uint8_t *, const uint64_t *, size_t, uint64_t *) { return DGPU_E_NODEVICE; }
int32_t dgpu_msm_g2_mont(const uint64_t *, const uint8_t *, const uint64_t *, size_t, uint64_t *) { return DGPU_E_NODEVICE; }
int32_t dgpu_bases_upload_g2(const uint64_t *, const uint8_t *, size_t, uint64_t *) { return DGPU_E_NODEVICE; }
int32_t dgpu_msm_g2_handle(uint64_t, size_t, const uint64_t *, size_t, int32_t, uint64_t *) { return DGPU_E_NODEVICE; }
int32_t dgpu_msm_g2_resident(uint64_t, size_t, uint64_t, size_t, size_t, uint64_t *) { return DGPU_E_NODEVICE; }
}
