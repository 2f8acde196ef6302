// torchscript_host.cpp - a libtorch host for a scripted molann_amd model (what an MD engine does with the
// file `torch.jit.script(model).save(...)` wrote, README.rst:49 of the reference).
//
//   torchscript_host <libmolann_torch.so> <model.pt> <frames.bin> <n_frames> <n_atoms> <out.bin> [--forces]
//
// frames.bin: n_frames*n_atoms*3 float32.  out.bin: the model output [n_frames, d_out] float32, followed
// with --forces by d(sum of outputs)/dx [n_frames, n_atoms, 3] (the quantity a biasing code needs).
// The operator library is loaded with dlopen before the model: that is all a C++ host has to add.
#include <dlfcn.h>
#include <torch/script.h>
#include <torch/csrc/autograd/autograd.h>

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s libmolann_torch.so model.pt frames.bin n_frames n_atoms out.bin [--forces]\n", argv[0]);
        return 2;
    }
    if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) {
        std::fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror());
        return 1;
    }
    const long n = std::stol(argv[4]), atoms = std::stol(argv[5]);
    const bool forces = argc > 7 && std::string(argv[7]) == "--forces";
    std::vector<float> host((size_t)(n * atoms * 3));
    std::ifstream in(argv[3], std::ios::binary);
    in.read(reinterpret_cast<char*>(host.data()), (std::streamsize)(host.size() * sizeof(float)));
    if (!in) {
        std::fprintf(stderr, "short read on %s\n", argv[3]);
        return 1;
    }
    try {
        torch::jit::script::Module model = torch::jit::load(argv[2], torch::kCUDA);
        model.eval();
        at::Tensor x = torch::from_blob(host.data(), {n, atoms, 3}, torch::kFloat32).to(torch::kCUDA);
        std::ofstream out(argv[6], std::ios::binary);
        auto dump = [&](const at::Tensor& t) {
            const at::Tensor c = t.detach().to(torch::kCPU).contiguous();
            out.write(reinterpret_cast<const char*>(c.data_ptr<float>()), (std::streamsize)(c.numel() * sizeof(float)));
        };
        if (forces) {
            x.requires_grad_(true);
            at::Tensor y = model.forward({x}).toTensor();
            dump(y);
            dump(torch::autograd::grad({y.sum()}, {x})[0]);
            std::printf("out %ld x %ld, forces %ld x %ld x 3\n", (long)y.size(0), (long)y.size(1), n, atoms);
        } else {
            torch::NoGradGuard no_grad;
            at::Tensor y = model.forward({x}).toTensor();
            dump(y);
            std::printf("out %ld x %ld\n", (long)y.size(0), (long)y.size(1));
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
