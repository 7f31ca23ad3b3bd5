// placeholder until the tiled edge-function kernel lands
#include "raster.h"
namespace mrt {
int raster_upload(RasterState&, const mrt_mesh*, int, hipStream_t, std::string&) { return MRT_OK; }
int raster_render(RasterState&, const RasterTextures&, int, int, const float*, uint32_t*, int, float*, int32_t*, hipStream_t,
                  std::string& err) { err = "rasterizer path not built yet"; return MRT_E_STATE; }
void raster_release(RasterState&) {}
}
