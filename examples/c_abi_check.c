/* Plain-C consumer of the drop-in boundary: includes the public header as C99, links libosmosis_hip.so and calls the
 * entry points that need no GPU (version / last-error / shape queries).  Built and run by tests/test_host_cpu.py. */
#include <stdio.h>
#include <string.h>

#include "osmosis_hip.h"

int main(void) {
  osm_conv_desc cd;
  osm_gemm_desc gd;
  osm_attn_desc ad;
  memset(&cd, 0, sizeof cd);
  memset(&gd, 0, sizeof gd);
  memset(&ad, 0, sizeof ad);
  const int v = osm_version();
  /* a null descriptor must be rejected with a status code and a message, not a crash */
  const int rc = osm_conv2d_nhwc(&cd, NULL);
  const char* msg = osm_last_error();
  printf("version %d.%d.%d rc %d msg %s\n", v >> 16, (v >> 8) & 255, v & 255, rc, msg ? msg : "(null)");
  printf("splitk_hint %d gn_nchunk %d attn_supported %d %d packed %lld\n", osm_splitk_hint(256, 1024, 1024, 9, 1),
         osm_gn_nchunk(65536), osm_attn_small_supported(64, 64), osm_attn_small_supported(1024, 64),
         osm_packed_weight_elems(256, 256, 3, 3, 0));
  return (rc == OSM_ERR_INVALID && msg && strlen(msg) > 0) ? 0 : 1;
}
