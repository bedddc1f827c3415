// Host pairing micro-benchmark: product Miller loop, final exponentiation, one Fq product -- with the flags given on the command line
// (g++ -O3 -std=c++17 [-mbmi2 -madx] -I rust-kzg-bn254_amd/csrc tools/ubench/host_pairing_time.cpp).  No GPU.
#include "host_pairing.h"
#include <chrono>
#include <cstdio>
using namespace kzg_host;
int main() {
    uint64_t k1[4] = {12345, 0, 0, 0}, k2[4] = {777, 0, 0, 0};
    G1 a = g1_mul_generator(k1), b = g1_mul_generator(k2);
    G2 q1 = g2_generator(), q2 = g2_mul_generator(k1);
    G1 ps[2] = {a, g1_neg(b)}; G2 qs[2] = {q1, q2};
    bool deg = false;
    auto t0 = std::chrono::steady_clock::now();
    Fq12 f;
    for (int i = 0; i < 50; ++i) f = miller_ate_product(ps, qs, 2, &deg);
    auto t1 = std::chrono::steady_clock::now();
    Fq12 g;
    for (int i = 0; i < 50; ++i) g = final_exponentiation_x(f);
    auto t2 = std::chrono::steady_clock::now();
    printf("miller product %.3f ms, final exp %.3f ms, one? %d\n", std::chrono::duration<double, std::milli>(t1 - t0).count() / 50,
           std::chrono::duration<double, std::milli>(t2 - t1).count() / 50, (int)fq12_is_one(g));
    auto t3 = std::chrono::steady_clock::now();
    Fq a1 = a.x, b1 = b.y, c;
    for (int i = 0; i < 1000000; ++i) { c = mul(a1, b1); a1 = c; }
    auto t4 = std::chrono::steady_clock::now();
    printf("fq mul %.1f ns (%llu)\n", std::chrono::duration<double, std::nano>(t4 - t3).count() / 1e6, (unsigned long long)a1.l[0]);
}
