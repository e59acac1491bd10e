// Error channel, version and the ONE table of environment switches for the C ABI (include/infodiff_hip.h).
#include "idf_common.h"
#include <stdarg.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void idf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* idf_last_error(void) { return g_err; }
extern "C" int idf_version(void) { return 100; }

// Every environment switch the library reads -- seven -- read ONCE, at first use (a C++11 magic static: initialised exactly once under
// concurrent first calls); everything else that used to be a getenv is a constant with its measurement in the comment beside it.
// The Python package has its own table (infodiffusion_amd/knobs.py); INTEGRATION.md section 5 lists both.
const IdfKnobs& idf_knobs() {
  static const IdfKnobs k = [] {
    IdfKnobs v;
    auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
    v.conv_rs = (int)num("IDF_CONV_RS", 1);             // 0: halo / direct-to-LDS kernels; 1: row-reuse form, two 256-thread workgroups per CU; 2: one of 512
    v.conv_rs_sync = (int)num("IDF_CONV_RS_SYNC", 1);   // the row-reuse data-gradient conv applies the GroupNorm backward itself (workgroups of an image meet at a counter)
    v.conv_ps = (int)num("IDF_CONV_PS", 1);             // persistent wave-specialised plain conv (bit 0: 3x3, bit 1: 1x1)
    v.conv_dlds_min = num("IDF_CONV_DLDS_MIN", 1536);   // workgroups from which a plain 256-pixel launch takes the direct-to-LDS form
    v.wgrad_kr3 = (int)num("IDF_WGRAD_KR3", 1);         // batched stride-1 3x3 weight gradient in the shared-tile form
    v.wgrad_tpb3 = (int)num("IDF_WGRAD_TPB3", 128);     // ... its pixel tiles per workgroup (profiles/r04_wgrad_tpb3.txt)
    v.wgrad_ring = (int)num("IDF_WGRAD_RING", 1);       // ... its 64- / 32-wide maps in the row-ring form (round 6: profiles/r06_wgrad.txt)
    return v;
  }();
  return k;
}
