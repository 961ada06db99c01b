// The matrix-core argmax kernels that also write the second-peak plane: ncc_mfma_kernels.hip once more, as a
// translation unit of its own (see there).
#define NCC_MF_WANT2 1
#include "ncc_mfma_kernels.hip"
