// gen_cloud.cpp -- synthetic uniform cloud used by the golden fixtures:
// libstdc++ std::uniform_real_distribution<float>(-L/2, L/2) on std::mt19937(seed),
// draw order x,y,z per particle.  Usage: gen_cloud <n> <seed> <L> <out.f32>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char** argv) {
    if (argc != 5) { std::fprintf(stderr, "usage: %s n seed L out\n", argv[0]); return 2; }
    long n = std::atol(argv[1]);
    unsigned seed = (unsigned)std::strtoul(argv[2], nullptr, 10);
    float L = (float)std::atof(argv[3]);
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> dist(-L / 2, L / 2);
    std::vector<float> v((size_t)n * 3);
    for (long i = 0; i < n * 3; i++) v[i] = dist(gen);
    FILE* f = std::fopen(argv[4], "wb");
    if (!f) return 1;
    std::fwrite(v.data(), sizeof(float), v.size(), f);
    std::fclose(f);
    return 0;
}
