// locator.h -- radar::Locator with the reference's signature (src/locate/locator.h:53-98), calling
// only the C-ABI of librmr.so.  Not thread-safe, one caller thread (as the reference).
#pragma once
#include <cstddef>
#include <iostream>
#include <vector>

#include "../rmr.h"
#include "detector.h"
#include "robot.h"
#include "views.h"

namespace radar {

class Locator {
   public:
    Locator() = delete;
    Locator(int image_width, int image_height, const Matx33f& intrinsic, const Matx44f& lidar_to_camera,
            const Matx44f& world_to_camera, float zoom_factor = 0.5f, std::size_t queue_size = 3,
            float min_depth_diff = 500, float max_depth_diff = 4000, float cluster_tolerance = 400,
            int min_cluster_size = 8, int max_cluster_size = 1000, float max_distance = 29300, int device = 0) {
        rmr_locator_cfg c;
        rmr_locator_cfg_default(&c);
        c.image_width = image_width, c.image_height = image_height;
        for (int i = 0; i < 9; ++i) c.intrinsic[i] = detail::mat_at(intrinsic, i);
        for (int i = 0; i < 16; ++i)
            c.lidar_to_camera[i] = detail::mat_at(lidar_to_camera, i), c.world_to_camera[i] = detail::mat_at(world_to_camera, i);
        c.zoom_factor = zoom_factor;
        c.queue_size = (int)queue_size;
        c.min_depth_diff = min_depth_diff, c.max_depth_diff = max_depth_diff;
        c.cluster_tolerance = cluster_tolerance;
        c.min_cluster_size = min_cluster_size, c.max_cluster_size = max_cluster_size;
        c.max_distance = max_distance;
        c.device = device;
        if (rmr_status s = rmr_locator_create(&c, &h_); s != RMR_OK) detail::throw_status(s);
    }
    ~Locator() { rmr_locator_destroy(h_); }
    Locator(const Locator&) = delete;
    Locator& operator=(const Locator&) = delete;

#ifdef RADAR_HAVE_PCL
    // locator.h:67, the reference's own signature
    void update(const pcl::PointCloud<pcl::PointXYZ>::Ptr& cloud) noexcept {
        if (!cloud) std::cerr << "cloud is null." << std::endl;
        else if (cloud->empty()) std::cerr << "cloud is empty." << std::endl;
        const CloudView v(cloud);
        detail::check_or_abort(rmr_locator_update(h_, v.xyz, v.size, v.stride_bytes, RMR_MEM_HOST));
    }
#endif
    // locate.cpp:158-220; a null / empty cloud prints the reference's message and returns
    void update(const CloudView& cloud) noexcept {
        if (!cloud.xyz && cloud.empty()) std::cerr << "cloud is null." << std::endl;
        else if (cloud.empty()) std::cerr << "cloud is empty." << std::endl;
        detail::check_or_abort(rmr_locator_update(h_, cloud.xyz, cloud.size, cloud.stride_bytes,
                                                  cloud.on_device ? RMR_MEM_DEVICE : RMR_MEM_HOST));
    }
    // locate.cpp:231-264
    void cluster() noexcept { detail::check_or_abort(rmr_locator_cluster(h_)); }
    // throughput mode (no reference counterpart): update + cluster of consecutive frames of this stream, kept as
    // frames 0 .. n-1 for a batched search; the cluster stage runs once over all frames.  The clouds share one
    // memory kind and point stride.
    void updateClusterBatch(const std::vector<CloudView>& clouds) noexcept {
        if (clouds.empty()) return;
        std::vector<const float*> ptr;
        std::vector<int> n;
        for (const CloudView& c : clouds) ptr.push_back(c.xyz), n.push_back(c.size);
        detail::check_or_abort(rmr_locator_update_cluster_batch(h_, ptr.data(), n.data(), clouds[0].stride_bytes,
                                                                clouds[0].on_device ? RMR_MEM_DEVICE : RMR_MEM_HOST,
                                                                (int)clouds.size()));
    }
    // locate.cpp:323-326
    void search(std::vector<Robot>& robots) const noexcept {
        if (robots.empty()) return;
        std::vector<rmr_robot> buf;
        buf.reserve(robots.size());
        for (const Robot& r : robots) buf.push_back(r.toC());
        detail::check_or_abort(rmr_locator_search(h_, buf.data(), (int)buf.size()));
        for (std::size_t i = 0; i < robots.size(); ++i)
            if (buf[i].has_location)
                robots[i].setLocationMetres(Point3f{buf[i].location[0], buf[i].location[1], buf[i].location[2]});
    }
    rmr_locator* handle() const { return h_; }

   private:
    rmr_locator* h_ = nullptr;
};

}  // namespace radar
