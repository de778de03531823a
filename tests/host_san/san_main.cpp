// san_main.cpp -- TEST INFRASTRUCTURE: drives the library's host code through its C ABI (include/mm3d.h) on the fake device
// layer: mm3d_estimate_maps_transforms on 1 and 16 streams (same bits required), both estimation methods, every descriptor
// type's table entry, the shard driver for a world of 3 ranks emulated in one process, the device-list driver (mm3d_create_devices: 1, 2 and 3
// fake devices and the duplicate-device hook, the pair records through the fake RCCL), composeMaps, the stage-by-stage
// calls, parameter parsing, the degenerate inputs of the reference's gtests and the error paths.  Built with
// -fsanitize=thread or -fsanitize=address,undefined by tests/host_san/build.sh; exit code 0 = the checks passed and no
// sanitizer spoke.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mm3d.h"

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("CHECK failed at line %d: %s\n", __LINE__, #x); ++failures; } } while (0)

struct Pt { float x, y, z; uint32_t rgba; };
static std::vector<Pt> make_cloud(int n, unsigned seed)
{
  std::vector<Pt> v(n);
  uint32_t s = seed * 2654435761u + 12345u;
  auto u = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (int i = 0; i < n; ++i) { v[i] = Pt{u() * 20.f + (float)seed, u() * 20.f, u() * 2.f, 0xff000000u | (s >> 8)}; }
  return v;
}

int main()
{
  setenv("MM3D_FAKE_DEVICES", "3", 1);                         // fake_hip.cpp: three "devices"
  const int kMaps = 7, kPts = 2400;
  std::vector<std::vector<Pt>> clouds;
  std::vector<mm3d_cloud_view> views;
  for (int m = 0; m < kMaps; ++m) clouds.push_back(make_cloud(kPts + 37 * m, (unsigned)m + 1));
  clouds.push_back({});                                        // an empty cloud among the clouds: counts as "no keypoints"
  for (auto &c : clouds) views.push_back(mm3d_cloud_view{c.data(), c.size(), sizeof(Pt), 12});
  const size_t n = views.size(), max_pairs = n * (n - 1) / 2;

  mm3d_ctx *ctx = nullptr;
  CHECK(mm3d_create(0, &ctx) == MM3D_OK && ctx);
  mm3d_params p;
  mm3d_params_default(&p);
  char buf[2048];
  CHECK(mm3d_params_to_string(&p, buf, sizeof(buf)) > 0);
  const char *argv[] = {"prog", "--descriptor_type", "FPFH", "--estimation_method", "SAC_IA", "--refine_transform", "1", "--max_iterations", "40"};
  CHECK(mm3d_params_from_command_line(9, argv, &p) == MM3D_OK && p.descriptor_type == MM3D_DESC_FPFH && p.estimation_method == MM3D_EST_SAC_IA);
  for (int d = 0; d < 6; ++d) CHECK(mm3d_descriptor_name(d) && mm3d_descriptor_dim(d) > 0 && mm3d_descriptor_from_string(mm3d_descriptor_name(d)) == d);
  CHECK(mm3d_descriptor_from_string("nope") == MM3D_EINVAL && mm3d_descriptor_name(9) == nullptr);

  // the whole job: one stream, then sixteen (C++ worker threads inside the library), both methods: the same bits
  for (int method = 0; method < 2; ++method) {
    p.estimation_method = method;
    p.descriptor_type = method ? MM3D_DESC_FPFH : MM3D_DESC_PFH;
    std::vector<float> T1(n * 16), T16(n * 16);
    std::vector<mm3d_pair_result> P1(max_pairs), P16(max_pairs);
    size_t n1 = 0, np1 = 0, n16 = 0, np16 = 0;
    CHECK(mm3d_set_streams(ctx, 1) == MM3D_OK);
    mm3d_srand(ctx, 1);
    CHECK(mm3d_estimate_maps_transforms(ctx, views.data(), n, &p, T1.data(), &n1, P1.data(), &np1) == MM3D_OK);
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(mm3d_set_streams(ctx, rep == 1 ? 5 : 16) == MM3D_OK);
      mm3d_srand(ctx, 1);
      CHECK(mm3d_estimate_maps_transforms(ctx, views.data(), n, &p, T16.data(), &n16, P16.data(), &np16) == MM3D_OK);
      CHECK(n1 == n16 && np1 == np16 && np1 == (size_t)kMaps * (kMaps - 1) / 2);
      CHECK(std::memcmp(T1.data(), T16.data(), n1 * 16 * sizeof(float)) == 0);
      for (size_t q = 0; q < np1; ++q)
        CHECK(P1[q].source_idx == P16[q].source_idx && P1[q].target_idx == P16[q].target_idx && std::memcmp(P1[q].transform, P16[q].transform, 64) == 0 &&
              P1[q].confidence == P16[q].confidence && P1[q].icp_iterations == P16[q].icp_iterations);
    }
    double fs = 0, ts = 0;
    CHECK(mm3d_last_run_stage_seconds(ctx, &fs, &ts) == MM3D_OK && ts >= fs);
    // the shard driver: a world of three ranks, one after the other in this process, merged like two all-gathers would
    const int world = 3;
    std::vector<mm3d_ctx *> rctx(world);
    std::vector<mm3d_shard *> sh(world);
    for (int r = 0; r < world; ++r) {
      CHECK(mm3d_create(0, &rctx[r]) == MM3D_OK);
      CHECK(mm3d_set_streams(rctx[r], 4) == MM3D_OK);
      mm3d_srand(rctx[r], 1);
      CHECK(mm3d_shard_begin(rctx[r], views.data(), n, &p, r, world, &sh[r]) == MM3D_OK);
    }
    std::vector<std::vector<unsigned char>> bundle(n);
    std::vector<uint64_t> npts(n, 0), nkp(n, 0);
    for (int r = 0; r < world; ++r) {
      std::vector<uint64_t> a(n), b(n);
      CHECK(mm3d_shard_bundle_sizes(sh[r], a.data(), b.data()) == MM3D_OK);
      for (size_t m = 0; m < n; ++m)
        if (mm3d_shard_map_owner(m, world) == r) {
          npts[m] = a[m]; nkp[m] = b[m];
          bundle[m].resize(mm3d_shard_bundle_bytes(a[m], b[m], p.descriptor_type) + 16);
          CHECK(mm3d_shard_pack(sh[r], m, bundle[m].data()) == MM3D_OK);
        }
    }
    std::vector<mm3d_pair_result> merged(max_pairs);
    std::vector<int> seen(max_pairs, 0);
    size_t npm = 0;
    for (int r = 0; r < world; ++r) {
      std::vector<size_t> maps; std::vector<const void *> srcs; std::vector<uint64_t> a, b;
      for (size_t m = 0; m < n; ++m)
        if (mm3d_shard_map_owner(m, world) != r) { maps.push_back(m); srcs.push_back(bundle[m].data()); a.push_back(npts[m]); b.push_back(nkp[m]); }
      CHECK(mm3d_shard_unpack_many(sh[r], maps.size(), maps.data(), srcs.data(), a.data(), b.data()) == MM3D_OK);
      std::vector<mm3d_pair_result> pr(max_pairs);
      std::vector<unsigned char> mine(max_pairs);
      size_t np = 0;
      CHECK(mm3d_shard_pairs(sh[r], pr.data(), mine.data(), max_pairs, &np) == MM3D_OK);
      npm = np;
      for (size_t q = 0; q < np; ++q) if (mine[q]) { merged[q] = pr[q]; ++seen[q]; }
    }
    CHECK(npm == np1);
    for (size_t q = 0; q < npm; ++q)
      CHECK(seen[q] == 1 && merged[q].source_idx == P1[q].source_idx && std::memcmp(merged[q].transform, P1[q].transform, 64) == 0 &&
            merged[q].confidence == P1[q].confidence);
    std::vector<float> Tg(n * 16);
    size_t ng = 0;
    CHECK(mm3d_global_transforms(merged.data(), npm, p.confidence_threshold, n, Tg.data(), &ng) == MM3D_OK);
    CHECK(ng == n1 && std::memcmp(Tg.data(), T1.data(), ng * 16 * sizeof(float)) == 0);
    for (int r = 0; r < world; ++r) { mm3d_shard_end(sh[r]); mm3d_destroy(rctx[r]); }
    // the same job behind the reference's one entry point on a DEVICE LIST (mm3d_create_devices): one, two and three fake
    // devices, a thread and a stream set per device inside the library, bundles pulled from the owner, the pair records
    // through the (fake) RCCL all-gather -- and the bits of one device; then two "devices" that are the same one (the test
    // hook: no communicator, records through host memory)
    for (int nd = 1; nd <= 4; ++nd) {
      const int list3[3] = {0, 1, 2}, dup[2] = {0, 0};
      const bool hook = nd == 4;
      mm3d_ctx *dc = nullptr;
      if (hook) {
        CHECK(mm3d_create_devices(dup, 2, &dc) == MM3D_EINVAL && dc == nullptr);        // not without the hook
        setenv("MM3D_DEVICES_ALLOW_DUPLICATES", "1", 1);
      }
      CHECK(mm3d_create_devices(hook ? dup : list3, hook ? 2 : nd, &dc) == MM3D_OK && dc);
      if (hook) unsetenv("MM3D_DEVICES_ALLOW_DUPLICATES");
      if (!dc) continue;
      CHECK(mm3d_device_count(dc) == (hook ? 2 : nd) && mm3d_device_at(dc, 0) == 0 && mm3d_devices_use_rccl(dc) == (hook ? 0 : 1));
      CHECK(mm3d_set_streams(dc, nd == 2 ? 1 : 4) == MM3D_OK);
      mm3d_set_debug(dc, 1);                                  // every device's gathered copy is read back and compared
      for (int rep = 0; rep < 2; ++rep) {
        std::vector<float> Td(n * 16);
        std::vector<mm3d_pair_result> Pd(max_pairs);
        size_t ndn = 0, npd = 0;
        mm3d_srand(dc, 1);
        CHECK(mm3d_estimate_maps_transforms(dc, views.data(), n, &p, Td.data(), &ndn, Pd.data(), &npd) == MM3D_OK);
        CHECK(ndn == n1 && npd == np1 && std::memcmp(Td.data(), T1.data(), n1 * 16 * sizeof(float)) == 0);
        for (size_t q = 0; q < np1 && q < npd; ++q)
          CHECK(Pd[q].source_idx == P1[q].source_idx && Pd[q].target_idx == P1[q].target_idx && std::memcmp(Pd[q].transform, P1[q].transform, 64) == 0 &&
                Pd[q].confidence == P1[q].confidence && Pd[q].icp_iterations == P1[q].icp_iterations);
        double ex = -1, ps = -1, gs = -1;
        CHECK(mm3d_last_run_device_seconds(dc, &ex, &ps, &gs) == MM3D_OK && ex >= 0 && ps >= ex && gs >= 0);
      }
      // a failing job on a device list leaves the context usable (every device's thread leaves through the barrier)
      mm3d_params bad = p;
      bad.descriptor_type = 17;
      std::vector<float> Tb(n * 16);
      size_t nb = 0, npb = 0;
      CHECK(mm3d_estimate_maps_transforms(dc, views.data(), n, &bad, Tb.data(), &nb, nullptr, &npb) != MM3D_OK && std::strlen(mm3d_last_error(dc)) > 0);
      mm3d_srand(dc, 1);
      CHECK(mm3d_estimate_maps_transforms(dc, views.data(), n, &p, Tb.data(), &nb, nullptr, &npb) == MM3D_OK && npb == np1 &&
            std::memcmp(Tb.data(), T1.data(), n1 * 16 * sizeof(float)) == 0);
      mm3d_destroy(dc);
    }
    {
      const int beyond[2] = {0, 7};
      mm3d_ctx *dc = nullptr;
      CHECK(mm3d_create_devices(beyond, 2, &dc) != MM3D_OK && dc == nullptr);          // no such device
      CHECK(mm3d_create_devices(nullptr, 1, &dc) == MM3D_EINVAL && mm3d_create_devices(beyond, 0, &dc) == MM3D_EINVAL);
    }
  }

  // many small maps (the pairs run in same-target batches) with a map that turns out to have no keypoints at the END of the
  // list (the workers assumed it had some: the pair loop is redone sequentially), on several stream / worker splits
  {
    std::vector<std::vector<Pt>> small;
    std::vector<mm3d_cloud_view> sv;
    for (int m = 0; m < 20; ++m) small.push_back(make_cloud(m == 19 ? 4 : 500 + 11 * m, 100u + (unsigned)m));
    for (auto &c : small) sv.push_back(mm3d_cloud_view{c.data(), c.size(), sizeof(Pt), 12});
    p.estimation_method = MM3D_EST_SAC_IA; p.descriptor_type = MM3D_DESC_FPFH;
    const size_t ns = sv.size(), mp = ns * (ns - 1) / 2;
    std::vector<float> Ta(ns * 16), Tb(ns * 16);
    std::vector<mm3d_pair_result> Pa(mp), Pb(mp);
    size_t na = 0, npa = 0, nb = 0, npb = 0;
    CHECK(mm3d_set_streams(ctx, 1) == MM3D_OK);
    mm3d_srand(ctx, 1);
    CHECK(mm3d_estimate_maps_transforms(ctx, sv.data(), ns, &p, Ta.data(), &na, Pa.data(), &npa) == MM3D_OK && npa == 19 * 18 / 2);
    const char *workers[] = {"3", "8", "1"};
    for (int rep = 0; rep < 3; ++rep) {
      setenv("MM3D_FEATURE_WORKERS", workers[rep], 1);
      CHECK(mm3d_set_streams(ctx, rep == 2 ? 3 : 8) == MM3D_OK);
      mm3d_srand(ctx, 1);
      CHECK(mm3d_estimate_maps_transforms(ctx, sv.data(), ns, &p, Tb.data(), &nb, Pb.data(), &npb) == MM3D_OK);
      CHECK(na == nb && npa == npb && std::memcmp(Ta.data(), Tb.data(), na * 16 * sizeof(float)) == 0);
      for (size_t q = 0; q < npa; ++q) CHECK(std::memcmp(Pa[q].transform, Pb[q].transform, 64) == 0 && Pa[q].confidence == Pb[q].confidence);
    }
    unsetenv("MM3D_FEATURE_WORKERS");
    // the same twenty maps on a list of three devices (round 6: one shared rand() table, per-map readiness): the map without
    // keypoints falsifies the table's assumption, the run falls back to the staged form -- same bits; then with that map
    // replaced by an ordinary one and map 7 published LATE by its owner (every other device has long pulled the rest and
    // waits for it, the table stops at its first use as a source): same bits as one device again; and the staged form by knob
    const int list3[3] = {0, 1, 2};
    mm3d_ctx *dc = nullptr;
    CHECK(mm3d_create_devices(list3, 3, &dc) == MM3D_OK && dc);
    if (dc) {
      CHECK(mm3d_set_streams(dc, 3) == MM3D_OK);
      mm3d_srand(dc, 1);
      CHECK(mm3d_estimate_maps_transforms(dc, sv.data(), ns, &p, Tb.data(), &nb, Pb.data(), &npb) == MM3D_OK);
      CHECK(na == nb && npa == npb && std::memcmp(Ta.data(), Tb.data(), na * 16 * sizeof(float)) == 0);
      for (size_t q = 0; q < npa && q < npb; ++q) CHECK(std::memcmp(Pa[q].transform, Pb[q].transform, 64) == 0 && Pa[q].confidence == Pb[q].confidence);
      small[19] = make_cloud(777, 555u);
      sv[19] = mm3d_cloud_view{small[19].data(), small[19].size(), sizeof(Pt), 12};
      CHECK(mm3d_set_streams(ctx, 2) == MM3D_OK);
      mm3d_srand(ctx, 1);
      CHECK(mm3d_estimate_maps_transforms(ctx, sv.data(), ns, &p, Ta.data(), &na, Pa.data(), &npa) == MM3D_OK && npa == 20 * 19 / 2);
      {
        mm3d_cloud *raw = nullptr, *down = nullptr;               // how many points map 7 has after the (fake) filters: the late knob's key
        CHECK(mm3d_cloud_create(ctx, small[7].data(), small[7].size(), sizeof(Pt), 12, &raw) == MM3D_OK);
        CHECK(mm3d_downsample(ctx, raw, p.resolution, &down) == MM3D_OK);
        char buf[32];
        std::snprintf(buf, sizeof buf, "%zu", mm3d_cloud_size(down));
        setenv("MM3D_FAKE_LATE_POINTS", buf, 1);
        mm3d_cloud_free(ctx, raw); mm3d_cloud_free(ctx, down);
      }
      mm3d_srand(dc, 1);
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(mm3d_estimate_maps_transforms(dc, sv.data(), ns, &p, Tb.data(), &nb, Pb.data(), &npb) == MM3D_OK);
        CHECK(na == nb && npa == npb && std::memcmp(Ta.data(), Tb.data(), na * 16 * sizeof(float)) == 0);
        for (size_t q = 0; q < npa && q < npb; ++q) CHECK(std::memcmp(Pa[q].transform, Pb[q].transform, 64) == 0 && Pa[q].confidence == Pb[q].confidence);
        // (no mm3d_srand before the second run on either side: both generators must stand where the sequential loop left them)
        if (rep == 0) {
          CHECK(mm3d_estimate_maps_transforms(ctx, sv.data(), ns, &p, Ta.data(), &na, Pa.data(), &npa) == MM3D_OK);
        }
      }
      unsetenv("MM3D_FAKE_LATE_POINTS");
      mm3d_destroy(dc);
    }
  }

  // degenerate inputs of the reference's gtests (R/test/test_map_merging.cpp:9-40) and the error paths
  {
    std::vector<float> T(n * 16);
    size_t no = 99, np = 99;
    CHECK(mm3d_estimate_maps_transforms(ctx, nullptr, 0, &p, T.data(), &no, nullptr, &np) == MM3D_OK && no == 0);
    CHECK(mm3d_estimate_maps_transforms(ctx, views.data(), 1, &p, T.data(), &no, nullptr, &np) == MM3D_OK && no == 1 && T[0] == 1.0f && T[5] == 1.0f);
    mm3d_params bad = p;
    bad.descriptor_type = 17;
    CHECK(mm3d_estimate_maps_transforms(ctx, views.data(), 3, &bad, T.data(), &no, nullptr, &np) != MM3D_OK && std::strlen(mm3d_last_error(ctx)) > 0);
    mm3d_cloud *none = nullptr;
    CHECK(mm3d_compose_maps(ctx, nullptr, 0, nullptr, 0, 0.05, &none) == MM3D_OK && none == nullptr);
    CHECK(mm3d_cloud_create(ctx, clouds[0].data(), clouds[0].size(), 7, 12, &none) != MM3D_OK);
  }
  // stage by stage, objects freed from another thread and after their context is gone (the pool is reference counted)
  {
    mm3d_ctx *c2 = nullptr;
    CHECK(mm3d_create(0, &c2) == MM3D_OK);
    mm3d_cloud *raw = nullptr, *down = nullptr, *filt = nullptr, *kp = nullptr;
    mm3d_normals *nrm = nullptr;
    mm3d_desc *desc = nullptr;
    CHECK(mm3d_cloud_create(c2, clouds[0].data(), clouds[0].size(), sizeof(Pt), 12, &raw) == MM3D_OK);
    CHECK(mm3d_downsample(c2, raw, 0.1, &down) == MM3D_OK && mm3d_remove_outliers(c2, down, 0.8, 50, &filt) == MM3D_OK);
    CHECK(mm3d_compute_normals(c2, filt, 0.6, &nrm) == MM3D_OK);
    CHECK(mm3d_detect_keypoints(c2, filt, nrm, MM3D_KP_SIFT, 5.0, 0.6, 0.1, &kp) == MM3D_OK);
    CHECK(mm3d_compute_descriptors(c2, filt, nrm, kp, MM3D_DESC_FPFH, 0.8, &desc) == MM3D_OK && mm3d_desc_dim(desc) == 33);
    std::vector<mm3d_corr> corr(mm3d_desc_size(desc) + 1);
    size_t nc = 0;
    CHECK(mm3d_find_correspondences(c2, desc, desc, 5, corr.data(), corr.size(), &nc) == MM3D_OK && nc > 0);
    float T[16]; size_t ninl = 0;
    CHECK(mm3d_estimate_transform_from_correspondences(c2, kp, kp, corr.data(), nc, 0.3, T, nullptr, 0, &ninl) == MM3D_OK);
    const mm3d_cloud *two[2] = {filt, down};
    float TT[32] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 2, 3, 1};
    mm3d_cloud *merged = nullptr;
    CHECK(mm3d_compose_maps(c2, two, 2, TT, 2, 0.05, &merged) == MM3D_OK && merged && mm3d_cloud_size(merged) > 0);
    mm3d_cloud *none2 = nullptr;
    CHECK(mm3d_compose_maps(c2, two, 2, TT, 1, 0.05, &none2) != MM3D_OK && none2 == nullptr);
    std::thread other([&] { mm3d_cloud_free(c2, raw); mm3d_cloud_free(c2, down); mm3d_normals_free(c2, nrm); });
    other.join();
    mm3d_destroy(c2);                                          // the objects below outlive their context
    mm3d_cloud_free(ctx, filt); mm3d_cloud_free(ctx, kp); mm3d_desc_free(ctx, desc); mm3d_cloud_free(ctx, merged);   // (through another context)
  }
  mm3d_destroy(ctx);
  std::printf(failures ? "FAILED: %d checks\n" : "host sanitizer driver ok (%d failed checks)\n", failures);
  return failures ? 1 : 0;
}
