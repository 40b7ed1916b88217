// gl3_prefill.hip — batched prefill (placeholder until the MFMA path lands in this file).
#include "gl3_ctx.h"

struct gl3_prefill_state { int unused; };

int32_t gl3_prefill_alloc(gl3_ctx* ctx) { (void)ctx; return GL3_OK; }
void gl3_prefill_free(gl3_ctx* ctx) { (void)ctx; }
int32_t gl3_prefill_run(gl3_ctx* ctx, const int32_t*, int32_t, int32_t) { GL3_FAIL(GL3_E_UNSUPPORTED, "batched prefill not built"); }
