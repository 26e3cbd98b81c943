// ccm_ctx.cpp -- context lifetime and error reporting for the C ABI (include/ccm_hot.h).
#include "ccm_internal.h"
#include <cstdarg>

int ccm_fail(ccm_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

hipStream_t ccm_aux_stream(ccm_ctx* c, int which)
{
    if (!c || which < 0 || which > 1) return nullptr;
    if (!c->aux[which]) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&c->aux[which], hipStreamNonBlocking, which == 0 ? least : 0) != hipSuccess) { c->aux[which] = nullptr; (void)hipGetLastError(); }
    }
    return c->aux[which];
}

extern "C" {

int ccm_abi_version(void) { return CCM_ABI_VERSION; }

ccm_ctx* ccm_create(int device, int flags)
{
    (void)flags;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    ccm_ctx* c = new (std::nothrow) ccm_ctx();
    if (!c) return nullptr;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    return c;
}

void ccm_destroy(ccm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_state_free(c);
    orb_state_free(c->orb);
    match_state_free(c->match);
    ba_state_free(c->ba);
    pose_state_free(c->pose);
    sim3_state_free(c->sim3);
    ess_state_free(c->ess);
    for (ProfLabel& L : c->prof) for (auto& e : L.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (hipStream_t& a : c->aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); a = nullptr; }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ccm_last_error(const ccm_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ccm_sync(ccm_ctx* c)
{
    if (!c) return CCM_E_ARG;
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

void* ccm_stream(ccm_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ccm_host_register(ccm_ctx* c, void* ptr, size_t bytes)
{
    if (!c || !ptr || bytes == 0) return c ? ccm_fail(c, CCM_E_ARG, "bad host buffer") : CCM_E_ARG;
    CCM_HIP(c, hipSetDevice(c->device));
    CCM_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return CCM_OK;
}

int ccm_host_unregister(ccm_ctx* c, void* ptr)
{
    if (!c || !ptr) return CCM_E_ARG;
    CCM_HIP(c, hipSetDevice(c->device));
    CCM_HIP(c, hipHostUnregister(ptr));
    return CCM_OK;
}

int ccm_profile_enable(ccm_ctx* c, int on)
{
    if (!c) return CCM_E_ARG;
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    c->prof_on = on != 0;
    for (ProfLabel& L : c->prof) L.used = 0;
    return CCM_OK;
}

int ccm_profile_read(ccm_ctx* c, float ms[CCM_PROF_COUNT], int32_t launches[CCM_PROF_COUNT])
{
    if (!c || !ms || !launches) return CCM_E_ARG;
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < CCM_PROF_COUNT; i++) {
        ProfLabel& L = c->prof[i];
        float sum = 0;
        for (size_t k = 0; k < L.used; k++) {
            float t = 0;
            if (hipEventElapsedTime(&t, L.ev[k].first, L.ev[k].second) == hipSuccess) sum += t;
        }
        ms[i] = sum; launches[i] = (int32_t)L.used; L.used = 0;
    }
    return CCM_OK;
}

}  // extern "C"
