// heffte_dump.cpp -- TEST INFRASTRUCTURE.  Golden-fixture generator: runs the heFFTe 2.1.0 *stock CPU* backend that is
// bundled with the reference (/root/reference/heffte/heffteBenchmark, compiled by oracle/Makefile) on a seed-4242 world
// exactly as heFFTe's own test does (test/test_fft3d.h:20-28: std::minstd_rand(4242), uniform_real(0,1), one draw per
// element, cast to complex) and writes input and forward output as raw fp64 (re,im) pairs.
//
//   heffte_dump S0 S1 S2 out_prefix      heFFTe index 0 is the FASTEST dimension: S0 = our N2, S1 = N1, S2 = N0.
//
// Output files: <prefix>.in, <prefix>.out (S0*S1*S2 complex each, heFFTe order = our [N0][N1][N2] order).
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "heffte.h"

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    if (argc != 5) {
        fprintf(stderr, "usage: %s S0 S1 S2 out_prefix\n", argv[0]);
        return 2;
    }
    const int         s0 = atoi(argv[1]), s1 = atoi(argv[2]), s2 = atoi(argv[3]);
    const std::string prefix = argv[4];
    heffte::box3d<>   world = {{0, 0, 0}, {s0 - 1, s1 - 1, s2 - 1}};

    std::minstd_rand                       park_miller(4242);
    std::uniform_real_distribution<double> unif(0.0, 1.0);
    std::vector<std::complex<double>>      input(world.count());
    for (auto& r : input) r = static_cast<std::complex<double>>(unif(park_miller));

    heffte::fft3d<heffte::backend::stock> fft(world, world, MPI_COMM_WORLD);
    std::vector<std::complex<double>>     output(fft.size_outbox());
    fft.forward(input.data(), output.data());

    auto dump = [](const std::string& path, const std::vector<std::complex<double>>& v) {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f || fwrite(v.data(), sizeof(std::complex<double>), v.size(), f) != v.size()) {
            fprintf(stderr, "cannot write %s\n", path.c_str());
            exit(3);
        }
        fclose(f);
    };
    dump(prefix + ".in", input);
    dump(prefix + ".out", output);
    MPI_Finalize();
    return 0;
}
