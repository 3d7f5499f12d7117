// ds_read_b128 bank-conflict microbenchmark for gfx950: which lanes of a wave64 are serviced together?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_conflict.hip -o /tmp/lds_conflict && /tmp/lds_conflict
// Every lane reads 16 bytes at a host-chosen LDS byte offset, 8 waves per CU, in a long unrolled loop; the time per
// instruction relative to the conflict-free pattern is the number of passes the LDS needed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void lds_read(const int *offs, int iters, int *out)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + offs[threadIdx.x & 63];
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        v4i r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[u]) : "v"(base), "n"(u * 4096));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345) out[0] = 1;
}

static double run(const char *name, std::function<int(int)> f, double base_ns)
{
    std::vector<int> h(64);
    for (int l = 0; l < 64; ++l) h[l] = f(l);
    int *d, *out;
    hipMalloc(&d, 256); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(lds_read, dim3(256), dim3(512), 0, 0, d, iters, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(lds_read, dim3(256), dim3(512), 0, 0, d, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 8 * 8);  // per wave-instruction per CU
    printf("%-58s %6.2f ns per ds_read_b128 per CU  (x%.2f)\n", name, ns, base_ns > 0 ? ns / base_ns : 1.0);
    hipFree(d); hipFree(out);
    return ns;
}

int main()
{
    const double b = run("linear: lane*16", [](int l) { return l * 16; }, 0);
    run("all lanes same address (broadcast)", [](int) { return 0; }, b);
    run("halves alias: (lane&31)*16 + (lane>>5)*1024", [](int l) { return (l & 31) * 16 + (l >> 5) * 1024; }, b);
    run("halves alias +128: (lane&31)*16 + (lane>>5)*(1024+128)", [](int l) { return (l & 31) * 16 + (l >> 5) * 1152; }, b);
    run("quarters alias: (lane&15)*16 + (lane>>4)*1024", [](int l) { return (l & 15) * 16 + (l >> 4) * 1024; }, b);
    run("eighths alias: (lane&7)*16 + (lane>>3)*1024", [](int l) { return (l & 7) * 16 + (l >> 3) * 1024; }, b);
    run("stride 32 B: lane*32", [](int l) { return l * 32; }, b);
    run("stride 64 B: lane*64", [](int l) { return l * 64; }, b);
    run("stride 256 B: lane*256 (all one bank group)", [](int l) { return l * 256; }, b);
    // the row-image pattern of conv_rows.hip for W = 13, RS = 16: pixel lj of 32, k-half kh = lane>>5 one piece on
    auto rows = [](int W, int rowb, int pieceb) {
        return [=](int l) { const int lj = l & 31, kh = l >> 5; const int r = lj / W, x = lj % W; return r * rowb + (x + 1) * 16 + kh * pieceb; };
    };
    run("rows W=13 rowb=1024 pieceb=256 (current)", rows(13, 1024, 256), b);
    run("rows W=13 rowb=1024+208 pieceb=256 (skewed)", rows(13, 1232, 256), b);
    run("rows W=13 rowb=1232, pieceb=256+128 (hypothetical)", rows(13, 1232, 384), b);
    run("rows W=26 rowb=2048 pieceb=512 (current)", rows(26, 2048, 512), b);
    run("rows W=26 rowb=2048+160 pieceb=512 (skewed)", rows(26, 2208, 512), b);
    run("rows W=52 rowb=4096 pieceb=1024", rows(52, 4096, 1024), b);
    // A-fragment pattern of conv_rows.hip: row = lj, ((row>>4)<<10) + ((row&15)<<4) + kh*256
    run("A frag: ((lj>>4)<<10) + ((lj&15)<<4) + kh*256", [](int l) { const int lj = l & 31, kh = l >> 5; return ((lj >> 4) << 10) + ((lj & 15) << 4) + kh * 256; }, b);
    run("A frag alt: lj*16 + kh*512", [](int l) { const int lj = l & 31, kh = l >> 5; return lj * 16 + kh * 512; }, b);
    return 0;
}
