#include <stdarg.h>
#include "common.h"
#include "../../include/seedhip.h"

namespace seedhip {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace seedhip

extern "C" const char* seedhip_last_error(void) { return seedhip::err_buf(); }
extern "C" int seedhip_abi_version(void) { return SEEDHIP_ABI_VERSION; }

// CRC32C (Castagnoli) of a host buffer: the checksum of TensorFlow's checkpoint files (tensorflow/core/lib/hash/crc32c.h),
// used by seed_rl_amd/tf_checkpoint.py when it reads / writes tf.train.Checkpoint bundles.  Host code, 8 tables.
namespace {
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
}  // namespace

extern "C" unsigned int seedhip_crc32c(const void* data, size_t n, unsigned int crc) {
  static const Crc32cTables tb;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = tb.t[7][lo & 0xFF] ^ tb.t[6][(lo >> 8) & 0xFF] ^ tb.t[5][(lo >> 16) & 0xFF] ^ tb.t[4][lo >> 24] ^
        tb.t[3][hi & 0xFF] ^ tb.t[2][(hi >> 8) & 0xFF] ^ tb.t[1][(hi >> 16) & 0xFF] ^ tb.t[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = tb.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}
