/* Native driver for the ASan/UBSan/LeakSanitizer run of the host library's CPU-only entry points
 * (tests/test_host.py::test_host_library_under_sanitizers).  TEST INFRASTRUCTURE ONLY.
 * argv[1] = directory of the test fixtures (tests/golden). */
#include <string>
#include <cstdio>
#include <cstdint>
#include <cstddef>
#include "livre_hip_driver.h"
int main( int argc, char** argv )
{
    const std::string g = argc > 1 ? argv[1] : ".";
    const std::string uvf = "uvf://" + g + "/mouse_reduced.uvf", bad = "uvf://" + g + "/nucleon.raw";
    int rc = lvh_selftest_cache(); std::printf("cache %d %s\n", rc, rc ? lvh_last_error() : "");
    rc = lvh_selftest_plugin_factory(); std::printf("factory %d %s\n", rc, rc ? lvh_last_error() : "");
    float m[4][16]; rc = lvh_selftest_camera(m); std::printf("camera %d\n", rc);
    rc = lvh_selftest_clip_planes(); std::printf("clip planes %d\n", rc);
    rc = lvh_selftest_renderer_parameters(); std::printf("renderer parameters %d\n", rc);
    size_t n = 0;
    rc = lvh_datasource_brick("raw:///nonexistent.raw#4,4,4,uint8", 0, nullptr, 0, &n); std::printf("raw missing rc=%d %s\n", rc, lvh_last_error());
    rc = lvh_datasource_brick("nosuch://x", 0, nullptr, 0, &n); std::printf("nosuch rc=%d %s\n", rc, lvh_last_error());
    rc = lvh_datasource_brick(uvf.c_str(), (1ull) | (0ull << 4), nullptr, 0, &n); std::printf("uvf brick rc=%d n=%zu\n", rc, n);
    rc = lvh_datasource_brick("mem://#64,64,64,16", 2, nullptr, 0, &n); std::printf("mem brick rc=%d n=%zu\n", rc, n);
    uint32_t v[3], mb[3], ov[3], rb[3], depth, dt, cc; float ws[3];
    rc = lvh_datasource_info(uvf.c_str(), v, mb, ov, ws, &depth, rb, &dt, &cc); std::printf("uvf info rc=%d depth %u\n", rc, depth);
    rc = lvh_datasource_info(bad.c_str(), v, mb, ov, ws, &depth, rb, &dt, &cc); std::printf("uvf bad file rc=%d %s\n", rc, lvh_last_error());
    std::printf( "DONE\n" );
    return 0;
}
