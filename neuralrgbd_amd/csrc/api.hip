// api.hip — library identification and error text of libnrgbd_hip.so.
#include "common.hpp"

extern "C" const char* nrgbd_version(void) { return "nrgbd_hip 0.5 (gfx950, CDNA4)"; }

extern "C" const char* nrgbd_strerror(int code) {
    switch (code) {
        case NRGBD_OK: return "success";
        case NRGBD_E_NULL: return "a required pointer is NULL";
        case NRGBD_E_SHAPE: return "a dimension is <= 0 or exceeds a kernel limit";
        case NRGBD_E_ALIGN: return "channel padding / pointer alignment violates the 16-byte texel rule";
        case NRGBD_E_ARG: return "an enum or flag argument is out of range";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown nrgbd error";
}
