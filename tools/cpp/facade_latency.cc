// Latency of ONE FBstabMpc::Solve through the C++11 facade (include/fbstab/fbstab_mpc.h),
// host pointers, Display::OFF - the call a user of the reference makes in a control loop
// (fbstab/fbstab_mpc.h:181-195).  bench.py writes the QP (the first instance of the
// BASELINE workload) to a file and runs this as a child process:
//   facade_latency <file> <repeats>
// file: int32 N nx nu nc, then the twelve sequences of FBstabMpc::ProblemData as doubles in
// member order (Q R S q r A B c E L d x0).  Prints one JSON object: the first (cold library)
// call apart, then the median / min of the repeats, each from a zero guess.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fbstab/fbstab_mpc.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  const int reps = atoi(argv[2]);
  int sz[4];
  if (fread(sz, sizeof(int), 4, f) != 4) return 4;
  const int N = sz[0], nx = sz[1], nu = sz[2], nc = sz[3];
  const int len[12] = {(N + 1) * nx * nx, (N + 1) * nu * nu, (N + 1) * nu * nx, (N + 1) * nx, (N + 1) * nu,
                       N * nx * nx, N * nx * nu, N * nx, (N + 1) * nc * nx, (N + 1) * nc * nu, (N + 1) * nc, nx};
  std::vector<std::vector<double>> a(12);
  for (int i = 0; i < 12; i++) {
    a[i].resize(len[i]);
    if (fread(a[i].data(), sizeof(double), len[i], f) != (size_t)len[i]) return 5;
  }
  fclose(f);
  using fbstab::FBstabMpc;
  using fbstab::MapMatrixSequence;
  FBstabMpc::ProblemDataRef qp;
  qp.Q = MapMatrixSequence(a[0].data(), N + 1, nx, nx);
  qp.R = MapMatrixSequence(a[1].data(), N + 1, nu, nu);
  qp.S = MapMatrixSequence(a[2].data(), N + 1, nu, nx);
  qp.q = MapMatrixSequence(a[3].data(), N + 1, nx, 1);
  qp.r = MapMatrixSequence(a[4].data(), N + 1, nu, 1);
  qp.A = MapMatrixSequence(a[5].data(), N, nx, nx);
  qp.B = MapMatrixSequence(a[6].data(), N, nx, nu);
  qp.c = MapMatrixSequence(a[7].data(), N, nx, 1);
  qp.E = MapMatrixSequence(a[8].data(), N + 1, nc, nx);
  qp.L = MapMatrixSequence(a[9].data(), N + 1, nc, nu);
  qp.d = MapMatrixSequence(a[10].data(), N + 1, nc, 1);
  qp.x0 = FBstabMpc::ConstVectorRef(a[11].data(), nx);
  FBstabMpc solver(N, nx, nu, nc);
  FBstabMpc::Options o = FBstabMpc::DefaultOptions();
  o.display_level = fbstab::Display::OFF;
  solver.UpdateOptions(o);
  std::vector<double> ms;
  int newton = 0, eflag = -1;
  for (int k = 0; k <= reps; k++) {
    FBstabMpc::Variable x(N, nx, nu, nc);
    const auto t0 = std::chrono::high_resolution_clock::now();
    const fbstab::SolverOut out = solver.Solve(qp, &x);
    const auto t1 = std::chrono::high_resolution_clock::now();
    ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
    newton = out.newton_iters;
    eflag = static_cast<int>(out.eflag);
  }
  const double first = ms[0];
  std::vector<double> rest(ms.begin() + 1, ms.end());
  std::sort(rest.begin(), rest.end());
  printf("{\"first_call_ms\": %.4f, \"median_ms\": %.4f, \"min_ms\": %.4f, \"repeats\": %d, \"newton_iters\": %d, \"eflag\": %d}\n",
         first, rest[rest.size() / 2], rest[0], reps, newton, eflag);
  return 0;
}
