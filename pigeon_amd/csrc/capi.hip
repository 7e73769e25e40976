// Version / error-string / workspace entry points of libpigeon_hip (see include/pigeon_hip.h).
#include "pgn_common.h"

extern "C" int pgn_version(void) { return PGN_VERSION; }

extern "C" const char* pgn_error_string(int code) {
    if (code == PGN_OK) return "ok";
    if (code == PGN_ERR_INVALID) return "invalid argument (null pointer, bad size/enum or misaligned buffer)";
    if (code == PGN_ERR_UNSUPPORTED) return "request not covered by the gfx950 kernels";
    if (code == PGN_ERR_SCRATCH) return "shape needs more scratch than the library's static device buffer holds";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown pigeon_hip error";
}

extern "C" size_t pgn_workspace_bytes(int op) {
    // Shape-independent scratch: none.  The two GroupNorm paths that reduce across workgroups size their caller-provided
    // workspace from the shape: pgn_groupnorm_workspace_bytes / pgn_groupnorm_nhwc_workspace_bytes.
    (void)op;
    return 0;
}
