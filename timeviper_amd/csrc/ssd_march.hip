// placeholder until the MFMA chunk-march kernel lands
#include "common.hpp"
bool tv_ssd_march_supported(int, int, int, int, int, int, int64_t, int64_t, int64_t, int64_t,
                            const void*, const void*, const void*, const void*) { return false; }
size_t tv_ssd_march_workspace_bytes(int, int, int, int, int, int) { return 0; }
int tv_ssd_march_launch(const void*, const void*, const void*, const void*, const void*,
                        const void*, const void*, const void*, void*, void*, void*, int, int, int,
                        int, int, int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                        int64_t, int64_t, int64_t, int64_t, int, int, float, float, int, void*,
                        size_t, hipStream_t) {
  TV_UNSUPPORTED("ssd_march: not built");
}
