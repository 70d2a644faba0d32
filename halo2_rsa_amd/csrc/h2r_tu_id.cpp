// libh2r.so, translation unit "id": the build ID -- SHA-256 of every file the library was compiled from (csrc/*, include/*),
// computed by halo2_rsa_amd/_build.py and passed as H2R_BUILD_ID_STR.  The marker in front lets a tool read it out of the file
// without loading the library (_build.lib_id).
#include "h2r.h"

#ifndef H2R_BUILD_ID_STR
#error "h2r_tu_id.cpp is compiled by halo2_rsa_amd/_build.py, which defines H2R_BUILD_ID_STR"
#endif

extern "C" const char *h2r_build_id(void) {
    static const char id[] = "H2R_BUILD_ID=" H2R_BUILD_ID_STR;
    return id + 13;
}
