// Diagnostic build of csrc/lstm.hip (-DLS_DIAG): where do the cycles of one forward step go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLS_DIAG -I include -I mod_extraction_amd/csrc \
//         tools/probe/lstm_probe.hip -o tools/probe/_bin/lstm_probe
#include "../../mod_extraction_amd/csrc/lstm.hip"
int g_mx_probe = 0;
#include <cstdio>
#include <vector>
int main()
{
    const int B = 128, T = 1024;
    std::vector<float> h((size_t)B * T), w(256 * 64), wi(512), bi(256), fc(64), one(1, 0.1f), st(B * 64, 0.f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t i = 0; i < w.size(); ++i) w[i] = ((float)((i * 40503u) % 2001) / 1000.f - 1.f) * 0.125f;
    for (size_t i = 0; i < wi.size(); ++i) wi[i] = ((float)((i * 9973u) % 2001) / 1000.f - 1.f) * 0.5f;
    for (size_t i = 0; i < bi.size(); ++i) bi[i] = 0.01f * (float)(i % 7);
    for (size_t i = 0; i < fc.size(); ++i) fc[i] = 0.05f * (float)(i % 5) - 0.1f;
    float *dx, *dl, *dw, *dwi, *db, *dfc, *dfb, *dh, *dc, *dy, *dst;
    hipMalloc(&dx, h.size() * 4); hipMalloc(&dl, h.size() * 4); hipMalloc(&dy, h.size() * 4);
    hipMalloc(&dw, w.size() * 4); hipMalloc(&dwi, wi.size() * 4); hipMalloc(&db, bi.size() * 4); hipMalloc(&dfc, 256); hipMalloc(&dfb, 4);
    hipMalloc(&dh, st.size() * 4); hipMalloc(&dc, st.size() * 4); hipMalloc(&dst, (size_t)B * T * 384 * 4);
    hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dl, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwi, wi.data(), wi.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bi.data(), bi.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dfc, fc.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dfb, one.data(), 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, st.data(), st.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dc, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int stash = 0; stash < 2; ++stash) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) mx_lstm_fwd(dx, T, dl, T, dwi, dw, db, db, dfc, dfb, dh, dc, dh, dc, dy, T, stash ? dst : nullptr, B, T, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("LS_ABL=%d stash=%d: %.1f ns per step\n", LS_ABL, stash, ms / 10 * 1e6 / T);
#ifdef LS_DIAG
        unsigned long long d[64];
        hipMemcpyFromSymbol(d, HIP_SYMBOL(ls_diag), sizeof(d));
        const char *names[5] = {"LDS reads landed", "16 packed FMAs", "reduce+act+exchange+cell", "h write landed", "barrier"};
        printf("forward, stash=%d: cycles per step (last 256-step block), wave 0 / wave 7\n", stash);
        double tot0 = 0, tot7 = 0;
        for (int i = 0; i < 5; ++i) { printf("  %-28s %7.1f %7.1f\n", names[i], d[i] / 256.0, d[56 + i] / 256.0); tot0 += d[i] / 256.0; tot7 += d[56 + i] / 256.0; }
        printf("  %-28s %7.1f %7.1f\n", "total", tot0, tot7);
#endif
    }
    return 0;
}
