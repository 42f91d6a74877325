// lab/top_launchers.h -- LAB BUILD ONLY: host-side launchers of lab/top_kernels.h (included by traversal.hip behind its shipped launchers).
#pragma once

// "top*": top-of-tree image per launch, then k_bvh2_top (SORTED: through the "sorted" mapping's permutation)
template <bool ANY, int LDS_N, int TOPN, int WAVES, bool SORTED = false, bool KEEP = false> void L_top(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    if (!s.top_image) {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!s.top_image) HIP_CHECK(hipMalloc(&s.top_image, kMaxTopNodes * sizeof(Node2)));
    }
    static_assert(TOPN <= kMaxTopNodes, "image buffer");
    // KEEP (measurement only): the image is reused while the array pointer and the image size stay the same -- shows what the
    // per-launch rebuild costs
    if (!KEEP || s.top_image_nodes != nodes || s.top_image_n != TOPN) {
        hipLaunchKernelGGL((k_bvh2_top_image<TOPN>), dim3(1), dim3(kWave), 0, stream, nodes, s.top_image);
        s.top_image_nodes = nodes; s.top_image_n = TOPN;
    }
    const int* perm = nullptr;
    if (SORTED) {
        ensure_sort_buffers(s, n);
        const int blocks = (n + kSortBlockRays - 1) / kSortBlockRays;
        hipLaunchKernelGGL(k_raysort_count, dim3(blocks), dim3(kSortThreads), 0, stream, nodes, rays, n, s.sort_keys, s.sort_totals);
        hipLaunchKernelGGL(k_raysort_scan, dim3(1), dim3(kSortCells), 0, stream, s.sort_totals, s.sort_totals + kSortCells);
        hipLaunchKernelGGL(k_raysort_scatter, dim3(blocks), dim3(kSortThreads), 0, stream, s.sort_keys, n, s.sort_totals + kSortCells,
            s.sort_perm);
        perm = s.sort_perm;
    }
    const int groups = (blocks_for(n) + WAVES - 1) / WAVES;
    hipLaunchKernelGGL((k_bvh2_top<ANY, LDS_N, 32, TOPN, WAVES>), dim3(groups), dim3(kWave * WAVES), 0, stream, nodes, tris, rays, hits, n,
        s.ctl(), s.deep_list, perm,
                       (const int4*)s.top_image);
    hipLaunchKernelGGL((k_bvh2_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list, s.deep_stack,
        (int*)nullptr);
}

template <bool ANY, int LDS_N, int TOPN, int WAVES, int REFILL> void L_top_refill_wpe(LAUNCH_ARGS) {
    ensure_deep_list(s, n);
    ensure_top_buffers(s);
    s.top_image_nodes = nullptr;
    const int groups = spill_checked(((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes, WAVES);
    ensure_spill(s, groups * WAVES);
    hipLaunchKernelGGL((k_bvh2_top_refill_wpe<ANY, LDS_N, TOPN, WAVES, REFILL>), dim3(groups), dim3(kWave * WAVES), 0, stream, nodes, tris,
        rays, hits, n, s.ctl(), s.deep_list,
                       (const int4*)s.top_image, s.tickets, mapped_node_ids(nodes), s.spill);
    hipLaunchKernelGGL((k_bvh2_top_finish<ANY>), dim3(1), dim3(kWave), 0, stream, nodes, tris, rays, hits, s.ctl(), s.deep_list,
        s.deep_stack, s.tickets, s.top_image, TOPN);
}

// "steal" (lab): whole chunks with work stealing inside the wave (k_bvh2_top_steal); small launches take the one-chunk kernel like the
// default
template <bool ANY, int LDS_N, int TOPN, int WAVES, int I0, int EVERY> void L_top_steal(LAUNCH_ARGS) {
    const int max_id = n < g_top_min_rays ? 0 : mapped_node_ids(nodes);
    if (max_id == 0) { L_single<ANY, 16, 32>(s, nodes, tris, rays, hits, n, stream); return; }
    ensure_deep_list(s, n);
    ensure_top_buffers(s);
    s.top_image_nodes = nullptr; s.order_rays = 0;
    const int groups = ((s.num_cus * (32 / WAVES) + kStripes - 1) / kStripes) * kStripes;
    hipLaunchKernelGGL((k_bvh2_top_steal<ANY, LDS_N, TOPN, WAVES, I0, EVERY>), dim3(groups), dim3(kWave * WAVES), 0, stream, nodes, tris,
        rays, hits, n, s.ctl(), s.deep_list,
                       s.top_image, s.tickets, max_id);
}
