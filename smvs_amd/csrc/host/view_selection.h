// Host mirror of smvs::ViewSelection (reference: lib/view_selection.h:20-45,
// lib/view_selection.cc:14-161): the neighbour views of a reference view,
// chosen by shared SfM features (with a bundle) or by camera pose (without).
// Scene-level, runs once per reference view on the host;
// views are the plain records the rest of the host mirror uses instead of
// mve::View.  A list entry the reference would hold as a null View::Ptr is a
// record with present == false.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

#include "image.h"

namespace smvs_amd {

class ViewSelection
{
public:
    struct Options  // lib/view_selection.h:23-28
    {
        std::size_t num_neighbors = 6;
        std::string embedding = "undistorted";
    };

    // what the selection reads of an mve::View
    struct ViewInfo
    {
        bool present = true;     // false: a hole in the scene's view list
        int id = 0;              // mve::View::get_id()
        CameraInfo cam;
        bool has_image = true;   // has_image(opts.embedding)
        int width = 0, height = 0;   // of that embedding
    };
    typedef std::vector<ViewInfo> ViewList;

public:
    ViewSelection(Options const& opts, ViewList const& views,
        Bundle::ConstPtr bundle = nullptr)
        : opts(opts), views(views), bundle(bundle) {}

    // Indices into the view list, best neighbour first.  Like the reference
    // (view_selection.cc:62, 91) the bundle-based selection identifies a view
    // by its id where it means its index: ids equal indices in MVE scenes.
    std::vector<std::size_t> get_neighbors_for_view(std::size_t view) const;

private:
    std::vector<std::size_t> bundle_based_selection(std::size_t view) const;
    std::vector<std::size_t> position_based_selection(std::size_t view) const;
    std::vector<std::size_t> get_sorted_neighbors(std::size_t view) const;

private:
    Options const& opts;
    ViewList const& views;
    Bundle::ConstPtr bundle;
};

} // namespace smvs_amd
