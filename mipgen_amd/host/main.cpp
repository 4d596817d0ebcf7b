// main.cpp — the drop-in `mipgen` command line: same options, inputs and output files as the reference
// (/root/reference/mipgen.cpp:2021-2037 main).  Everything is in libmipgen_host.so (options, input stage, selection stage, the
// accelerated tile_regions driver) and libmipgen_accel.so (HIP, gfx950); this file only wires them together.
#include <cstdio>
#include <iostream>

#include "../../include/mipgen_host.h"

int main(int argc, char** argv)
{
    mipgen_design* d = nullptr;
    int rc = mipgen_design_open(argc, argv, &d);
    if (rc) {
        if (rc == MIPGEN_HOST_E_USAGE && mipgen_host_last_circumstance() == 1) std::cerr << mipgen_host_last_error() << std::endl;   // usage / option errors (mipgen.cpp:140-145)
        else std::cerr << mipgen_host_last_error() << std::endl;                                                                  // mipgen.cpp:2029-2032
        // a std::exception (a bad integer, a BED line with two fields, an option without its value ...) is reported and the reference's main() then falls
        // off its end: exit status 0 (mipgen.cpp:2033-2036) - reproduced; `throw <int>` paths exit with 1 (:2029-2032)
        if (mipgen_host_last_circumstance() == -1) return 0;                  // (-2 = a std::exception the reference cannot raise: this port's own failure, exit status 1)
        return 1;
    }
    rc = mipgen_design_run(d, 0);
    const int circ = rc ? mipgen_host_last_circumstance() : 0;
    const bool exception_path = circ == -1;                                              // a std::exception the reference raises too: as above
    if (circ == -1 || circ == -2) std::cerr << mipgen_host_last_error() << std::endl;
    else if (rc) std::cerr << "unable to tile sequences due to circumstance " << mipgen_host_last_circumstance() << std::endl;
    mipgen_design_close(d);
    return rc && !exception_path ? 1 : 0;
}
