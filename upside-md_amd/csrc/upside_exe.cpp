// `upside_hip`: the executable face of libupside_hip.so -- what /root/reference/src/main.cpp:756-759 is to libupside.so.
// One process per GPU under a launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (main_cli.cpp), or a single process.
#include "../../include/upside_engine_c.h"
int main(int argc, const char* const* argv) { return upside_main(argc, argv, 1); }
