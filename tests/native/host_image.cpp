// Host-side check of mrt_host::Surface's overlay text and PNG writer (raytracing_amd/host/mrt_host.hpp): draws the
// reference's two overlay lines (game.cpp:108-112) into a small frame and writes it as a PNG.  Needs no GPU.
// usage: host_image out.png width height
#include <cstdio>
#include <cstdlib>

#include "../../raytracing_amd/host/mrt_host.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    mrt_host::Surface s(atoi(argv[2]), atoi(argv[3]));
    s.Clear(0x00102030);
    s.PrintFrameStats(0.001234f, 810);
    s.Print("AZ az 09 ?", 2, 22, 0x00ffffff);          // upper case draws as lower case; an unknown character is a blank
    s.Print("clipped", s.GetWidth() - 9, s.GetHeight() - 3, 0x0000ff00);   // runs over the right and bottom edges
    return s.SavePNG(argv[1]) ? 0 : 1;
}
