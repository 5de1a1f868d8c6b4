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


// A stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): the closed serving loop gives
// central inference and the train step DISJOINT sets of CUs, so that an inference kernel never waits for the persistent
// workgroups of a train kernel to retire (learner_server.LearnerServer, SEEDRL_CU_SPLIT).
// mask: `words` 32-bit words, bit i of word w = compute unit 32 w + i in the runtime's enumeration.
extern "C" int seedhip_stream_create_cu_mask(const unsigned int* mask, int words, void** stream) {
  SEEDHIP_REQUIRE(mask && words >= 1 && stream, "stream_create_cu_mask: null pointer / no mask words");
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) return seedhip::fail(SEEDHIP_ERR_LAUNCH, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream = (void*)s;
  return SEEDHIP_OK;
}

extern "C" int seedhip_stream_destroy(void* stream) {
  if (!stream) return SEEDHIP_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return seedhip::fail(SEEDHIP_ERR_LAUNCH, "hipStreamDestroy: %s", hipGetErrorString(e));
  return SEEDHIP_OK;
}
