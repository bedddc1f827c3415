// Host pairing micro-benchmark: product Miller loop, final exponentiation, one Fq product -- with the flags given on the command line
// (g++ -O3 -std=c++17 [-mbmi2 -madx] -I rust-kzg-bn254_amd/csrc tools/ubench/host_pairing_time.cpp).  No GPU.
#include "host_pairing.h"
#include <chrono>
#include <cstdio>
#include <thread>
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
    {   // the two Miller loops on two threads (one std::thread per call) against the shared loop above
        auto u0 = std::chrono::steady_clock::now();
        Fq12 h;
        for (int i = 0; i < 50; ++i) {
            Fq12 f1, f2;
            bool d1 = false, d2 = false;
            std::thread th([&] { f2 = miller_ate_product(ps + 1, qs + 1, 1, &d2); });
            f1 = miller_ate_product(ps, qs, 1, &d1);
            th.join();
            h = mul(f1, f2);
        }
        auto u1 = std::chrono::steady_clock::now();
        printf("two threads: %.3f ms, equal to the shared loop? %d\n", std::chrono::duration<double, std::milli>(u1 - u0).count() / 50, (int)(memcmp(&h, &f, sizeof f) == 0));
    }
    auto t3 = std::chrono::steady_clock::now();
    Fq a1 = a.x, b1 = b.y, c;
    for (int i = 0; i < 1000000; ++i) { c = mul(a1, b1); a1 = c; }
    auto t4 = std::chrono::steady_clock::now();
    printf("fq mul %.1f ns (%llu)\n", std::chrono::duration<double, std::nano>(t4 - t3).count() / 1e6, (unsigned long long)a1.l[0]);
}
