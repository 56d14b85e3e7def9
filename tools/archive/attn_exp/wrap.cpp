// Experiment wrapper: attention_flash.hip built stand-alone with -DAP_PIPE_ABL=<mask> (tools/attn_ablate.py)
#include "../../atlaspatch_amd/csrc/ap_common.h"
#include <cstdio>
namespace ap {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
}
extern "C" int attn_exp(int dtype, const void* qkv, void* out, int n, int tokens, int heads, void* stream) {
    return ap::launch_attention_flash(dtype, qkv, out, n, tokens, heads, 64, 0.125f, (hipStream_t)stream);
}
