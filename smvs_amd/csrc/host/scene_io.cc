#include "scene_io.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <future>
#include <sstream>
#include <stdexcept>

#include "depth_optimizer.h"
#include "png_io.h"
#include "jpeg_io.h"
#include "sgm_stereo.h"
#include "stereo_view.h"
#include "view_queue.h"

namespace smvs_amd {

namespace {

char const MVEI_SIGNATURE[] = "\211MVE_IMAGE\n";   // 11 bytes
int const MVEI_SIGNATURE_LEN = 11;
// mve::ImageType: UNKNOWN, UINT8, UINT16, UINT32, UINT64, SINT8, SINT16,
// SINT32, SINT64, FLOAT, DOUBLE
int const MVEI_UINT8 = 1, MVEI_FLOAT = 9;

bool
file_exists(std::string const& path)
{
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}

void
read_header(std::ifstream& in, std::string const& path, int32_t* whct)
{
    char sig[MVEI_SIGNATURE_LEN];
    in.read(sig, MVEI_SIGNATURE_LEN);
    if (!in || std::memcmp(sig, MVEI_SIGNATURE, MVEI_SIGNATURE_LEN) != 0)
        throw std::runtime_error("not an .mvei file: " + path);
    in.read(reinterpret_cast<char*>(whct), 4 * sizeof(int32_t));
    if (!in || whct[0] <= 0 || whct[1] <= 0 || whct[2] <= 0)
        throw std::runtime_error("bad .mvei header: " + path);
}

template <typename T>
typename Image<T>::Ptr
load_mvei(std::string const& path, int type)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    int32_t whct[4];
    read_header(in, path, whct);
    if (whct[3] != type)
        throw std::runtime_error("unexpected .mvei pixel type in " + path);
    typename Image<T>::Ptr img = Image<T>::create(whct[0], whct[1], whct[2]);
    std::size_t const bytes = (std::size_t)whct[0] * whct[1] * whct[2] * sizeof(T);
    in.read(reinterpret_cast<char*>(img->begin()), (std::streamsize)bytes);
    if (!in)
        throw std::runtime_error("truncated .mvei file: " + path);
    return img;
}

template <typename T>
void
save_mvei_typed(std::string const& path, Image<T> const& img, int type)
{
    std::ofstream out(path.c_str(), std::ios::binary);
    if (!out)
        throw std::runtime_error("cannot write " + path);
    out.write(MVEI_SIGNATURE, MVEI_SIGNATURE_LEN);
    int32_t const whct[4] = { img.width(), img.height(), img.channels(), type };
    out.write(reinterpret_cast<char const*>(whct), sizeof(whct));
    out.write(reinterpret_cast<char const*>(img.begin()),
        (std::streamsize)((std::size_t)img.width() * img.height() * img.channels()
            * sizeof(T)));
    if (!out)
        throw std::runtime_error("write failed: " + path);
}

std::string
trim(std::string const& s)
{
    std::size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos)
        return "";
    std::size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

// meta.ini of an MVE view
void
parse_meta_ini(std::string const& path, SceneView* view)
{
    std::ifstream in(path.c_str());
    if (!in)
        throw std::runtime_error("cannot open " + path);
    std::string line, section;
    view->camera = CameraInfo();
    view->camera.flen = 0.0f;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#')
            continue;
        if (line[0] == '[') {
            section = line.substr(1, line.find(']') - 1);
            continue;
        }
        std::size_t const eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        std::string const key = section + "." + trim(line.substr(0, eq));
        std::istringstream val(line.substr(eq + 1));
        if (key == "camera.focal_length")
            val >> view->camera.flen;
        else if (key == "camera.pixel_aspect")
            val >> view->camera.paspect;
        else if (key == "camera.principal_point")
            val >> view->camera.ppoint[0] >> view->camera.ppoint[1];
        else if (key == "camera.rotation")
            for (int i = 0; i < 9; ++i)
                val >> view->camera.rot[i];
        else if (key == "camera.translation")
            for (int i = 0; i < 3; ++i)
                val >> view->camera.trans[i];
        else if (key == "view.id")
            val >> view->id;
        else if (key == "view.name")
            val >> view->name;
    }
}

} // namespace

ByteImage::Ptr
load_mvei_u8(std::string const& path)
{
    return load_mvei<uint8_t>(path, MVEI_UINT8);
}

FloatImage::Ptr
load_mvei_float(std::string const& path)
{
    return load_mvei<float>(path, MVEI_FLOAT);
}

void
save_mvei(std::string const& path, ByteImage::ConstPtr image)
{
    save_mvei_typed<uint8_t>(path, *image, MVEI_UINT8);
}

void
save_mvei(std::string const& path, FloatImage::ConstPtr image)
{
    save_mvei_typed<float>(path, *image, MVEI_FLOAT);
}

bool
mvei_header(std::string const& path, int* whct)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in)
        return false;
    int32_t h[4];
    try {
        read_header(in, path, h);
    } catch (std::exception const&) {
        return false;
    }
    for (int i = 0; i < 4; ++i)
        whct[i] = h[i];
    return true;
}

bool
SceneView::has_image(std::string const& embedding) const
{
    return present && (file_exists(directory + "/" + embedding + ".mvei")
        || file_exists(directory + "/" + embedding + ".png")
        || file_exists(directory + "/" + embedding + ".jpg")
        || file_exists(directory + "/" + embedding + ".jpeg"));
}

std::string
SceneView::image_path(std::string const& embedding) const
{
    // (.mvei, then .png, then .jpg -- makescene keeps a camera's JPEG as
    // original.jpg, smvsrecon --image=original reads it, app/smvsrecon.cc:41, 156)
    std::string const base = directory + "/" + embedding;
    if (!file_exists(base + ".mvei") && file_exists(base + ".png"))
        return base + ".png";
    if (!file_exists(base + ".mvei") && file_exists(base + ".jpg"))
        return base + ".jpg";
    if (!file_exists(base + ".mvei") && file_exists(base + ".jpeg"))
        return base + ".jpeg";
    return base + ".mvei";
}

ByteImage::Ptr
SceneView::load_byte_image(std::string const& embedding) const
{
    std::string const path = image_path(embedding);
    if (path.size() > 4 && path.substr(path.size() - 4) == ".png")
        return load_png_u8(path);
    if ((path.size() > 4 && path.substr(path.size() - 4) == ".jpg")
        || (path.size() > 5 && path.substr(path.size() - 5) == ".jpeg"))
        return load_jpeg_u8(path);
    return load_mvei_u8(path);
}

bool
SceneView::image_size(std::string const& embedding, int* whc) const
{
    std::string const path = image_path(embedding);
    if (path.size() > 4 && path.substr(path.size() - 4) == ".png")
        return png_header(path, whc);
    if ((path.size() > 4 && path.substr(path.size() - 4) == ".jpg")
        || (path.size() > 5 && path.substr(path.size() - 5) == ".jpeg"))
        return jpeg_header(path, whc);
    int whct[4] = { 0, 0, 0, 0 };
    if (!mvei_header(path, whct))
        return false;
    whc[0] = whct[0];
    whc[1] = whct[1];
    whc[2] = whct[2];
    return true;
}

ByteImage::Ptr
rescale_half_size_gaussian(ByteImage::ConstPtr img)
{
    // mve::image::rescale_half_size_gaussian<uint8_t>(img, sigma2 = 0.75f)
    // [MVE-unverified M29]: output ((w + 1) / 2, (h + 1) / 2); every output
    // pixel accumulates the 4 x 4 input window around (2x, 2y) .. (2x+1, 2y+1)
    // (indices clamped to the image) with the weights exp(-d^2 / (2 sigma^2))
    // of the three distances of a 4 x 4 window's cells from its centre, in a
    // float accumulator (math::Accum<unsigned char>), normalised by the sum of
    // the weights and rounded half away from zero.
    int const iw = img->width(), ih = img->height(), ic = img->channels();
    if (iw < 2 || ih < 2)
        throw std::invalid_argument("rescale_half_size_gaussian: image too small");
    int const ow = (iw + 1) >> 1, oh = (ih + 1) >> 1;
    ByteImage::Ptr out = ByteImage::create_for_overwrite(ow, oh, ic);
    float const sigma2 = 0.75f;
    float const w1 = std::exp(-0.5f / (2.0f * sigma2));
    float const w2 = std::exp(-2.5f / (2.0f * sigma2));
    float const w3 = std::exp(-4.5f / (2.0f * sigma2));
    float const wrow[4][4] = { { w3, w2, w2, w3 }, { w2, w1, w1, w2 },
        { w2, w1, w1, w2 }, { w3, w2, w2, w3 } };
    uint8_t const* src = img->begin();
    uint8_t* dst = out->begin();
    std::size_t const rowstride = (std::size_t)iw * ic;
    for (int y = 0; y < oh; ++y) {
        int const y2 = y << 1;
        uint8_t const* row[4] = { src + (std::size_t)std::max(0, y2 - 1) * rowstride,
            src + (std::size_t)y2 * rowstride,
            src + (std::size_t)std::min(ih - 1, y2 + 1) * rowstride,
            src + (std::size_t)std::min(ih - 1, y2 + 2) * rowstride };
        for (int x = 0; x < ow; ++x) {
            int const x2 = x << 1;
            int const xi[4] = { std::max(0, x2 - 1) * ic, x2 * ic,
                std::min(iw - 1, x2 + 1) * ic, std::min(iw - 1, x2 + 2) * ic };
            for (int c = 0; c < ic; ++c) {
                float v = 0.0f, w = 0.0f;
                for (int r = 0; r < 4; ++r)
                    for (int k = 0; k < 4; ++k) {
                        v += (float)row[r][xi[k] + c] * wrow[r][k];
                        w += wrow[r][k];
                    }
                float const q = v / w;
                *dst++ = (uint8_t)(q > 0.0f ? std::floor(q + 0.5f) : std::ceil(q - 0.5f));
            }
        }
    }
    return out;
}

Scene::Ptr
Scene::create(std::string const& path)
{
    Ptr scene(new Scene());
    scene->path = path;
    std::string const views_dir = path + "/views";
    DIR* dir = ::opendir(views_dir.c_str());
    if (dir == nullptr)
        throw std::runtime_error("cannot open scene directory " + views_dir);
    std::vector<SceneView> found;
    while (struct dirent* entry = ::readdir(dir)) {
        std::string const name = entry->d_name;
        if (name.size() < 5 || name.substr(name.size() - 4) != ".mve")
            continue;
        SceneView v;
        v.directory = views_dir + "/" + name;
        if (!file_exists(v.directory + "/meta.ini"))
            continue;
        parse_meta_ini(v.directory + "/meta.ini", &v);
        v.present = true;
        found.push_back(v);
    }
    ::closedir(dir);
    // list index = view id, holes are null views (mve::Scene::init_views)
    int max_id = -1;
    for (auto const& v : found)
        max_id = std::max(max_id, v.id);
    scene->views.assign((std::size_t)(max_id + 1), SceneView());
    for (auto const& v : found) {
        if (v.id < 0)
            throw std::runtime_error("negative view id in " + v.directory);
        scene->views[(std::size_t)v.id] = v;
    }
    return scene;
}

Bundle::Ptr
load_mve_bundle(std::string const& path)
{
    std::ifstream in(path.c_str());
    if (!in)
        throw std::runtime_error("cannot open bundle " + path);
    std::string magic, version;
    in >> magic >> version;
    if (magic != "drews" || version != "1.0")
        throw std::runtime_error("not an MVE bundle (drews 1.0): " + path);
    int num_cameras = 0, num_features = 0;
    in >> num_cameras >> num_features;
    if (!in || num_cameras < 0 || num_features < 0)
        throw std::runtime_error("bad bundle header: " + path);
    // cameras: focal k1 k2 / 9 rotation / 3 translation (the views' meta.ini
    // hold the same cameras; only the features are kept)
    for (int c = 0; c < num_cameras; ++c) {
        double skip;
        for (int i = 0; i < 15; ++i)
            in >> skip;
    }
    Bundle::Ptr bundle(new Bundle());
    bundle->features.resize((std::size_t)num_features);
    for (int f = 0; f < num_features; ++f) {
        Bundle::Feature3D& feat = bundle->features[(std::size_t)f];
        int rgb[3], num_refs = 0;
        in >> feat.pos[0] >> feat.pos[1] >> feat.pos[2];
        in >> rgb[0] >> rgb[1] >> rgb[2] >> num_refs;
        for (int r = 0; r < num_refs; ++r) {
            int view_id = 0, feature_id = 0;
            double error = 0.0;
            in >> view_id >> feature_id >> error;
            feat.view_ids.push_back(view_id);
        }
    }
    if (!in)
        throw std::runtime_error("truncated bundle: " + path);
    return bundle;
}

Bundle::Ptr
Scene::get_bundle(void) const
{
    return load_mve_bundle(path + "/synth_0.out");
}

ReconReport
reconstruct_scene(std::string const& scene_path, ReconSettings const& conf_in)
{
    typedef std::chrono::steady_clock Clock;
    ReconSettings conf = conf_in;
    Scene::Ptr scene = Scene::create(scene_path);
    std::vector<SceneView>& views = scene->get_views();

    // app/smvsrecon.cc:404-421
    Bundle::Ptr bundle;
    try {
        bundle = scene->get_bundle();
    } catch (std::exception const&) {
        bundle = nullptr;
        conf.use_sgm = true;
        if (conf.sgm_max == 0.0f)
            throw std::invalid_argument("no bundle file and no SGM depth range "
                "(sgm_min / sgm_max)");
    }
    // :423-427
    if (conf.view_ids.empty())
        for (auto const& v : views)
            if (v.present && v.is_camera_valid())
                conf.view_ids.push_back(v.id);

    // automatic input scale, :477-500: the average pixel count of the views
    // that hold the embedding against --max-pixels
    if (conf.input_scale < 0) {
        double avg_image_size = 0.0;
        int view_counter = 0;
        for (auto const& v : views) {
            int whc[3] = { 0, 0, 0 };
            if (!v.present || !v.has_image(conf.image_embedding)
                || !v.image_size(conf.image_embedding, whc))
                continue;
            avg_image_size += (double)((uint32_t)whc[0] * (uint32_t)whc[1]);
            view_counter += 1;
        }
        conf.input_scale = 0;
        if (view_counter > 0) {
            avg_image_size /= (double)view_counter;
            if (avg_image_size > (double)conf.max_pixels)
                conf.input_scale = (int)std::ceil(std::log2(
                    avg_image_size / (double)conf.max_pixels) / 2);
        }
    }
    std::string const input_name = conf.input_scale > 0                // :502-508
        ? "undist-L" + std::to_string(conf.input_scale) : conf.image_embedding;
    ReconReport report;
    report.input_name = input_name;
    report.input_scale = conf.input_scale;
    report.output_name = std::string(conf.use_shading ? "smvs-S" : "smvs-B")
        + std::to_string(conf.input_scale);                           // :510-515

    // :517-555
    std::vector<int> reconstruction_list;
    for (int id : conf.view_ids) {
        if (id < 0 || id >= (int)views.size() || !views[(std::size_t)id].present
            || !views[(std::size_t)id].has_image(conf.image_embedding))
            continue;
        if (conf.force_recon || !views[(std::size_t)id].has_image(report.output_name))
            reconstruction_list.push_back(id);
        else
            report.already_done.push_back(id);
    }

    // view selection, :560-612
    ViewSelection::ViewList infos(views.size());
    for (std::size_t i = 0; i < views.size(); ++i) {
        SceneView const& v = views[i];
        ViewSelection::ViewInfo& info = infos[i];
        info.present = v.present;
        info.id = v.id;
        info.cam = v.camera;
        int whc[3] = { 0, 0, 0 };
        info.has_image = v.present && v.has_image(conf.image_embedding)
            && v.image_size(conf.image_embedding, whc);
        info.width = whc[0];
        info.height = whc[1];
    }
    ViewSelection::Options select_opts;
    select_opts.num_neighbors = conf.num_neighbors;
    select_opts.embedding = conf.image_embedding;
    ViewSelection selection(select_opts, infos, bundle);
    std::vector<std::vector<std::size_t>> neighbors;
    std::vector<int> final_list;
    for (int id : reconstruction_list) {
        std::vector<std::size_t> nb = selection.get_neighbors_for_view((std::size_t)id);
        if (nb.size() < conf.min_neighbors) {
            report.skipped_few_neighbors.push_back(id);
            continue;
        }
        final_list.push_back(id);
        neighbors.push_back(nb);
    }

    // the input embedding at the input scale, :621-650: created once for every
    // view a task will read and saved into the view directory (as the PNG
    // mve::View::save_view writes for a byte image)
    if (conf.input_scale > 0) {
        std::vector<char> needed(views.size(), 0);
        for (std::size_t v = 0; v < final_list.size(); ++v) {
            needed[(std::size_t)final_list[v]] = 1;
            for (std::size_t n : neighbors[v])
                needed[n] = 1;
        }
        std::vector<std::future<void>> resize_tasks;
        for (std::size_t i = 0; i < views.size(); ++i) {
            SceneView const& view = views[i];
            if (!needed[i] || !view.present || !view.has_image(conf.image_embedding)
                || view.has_image(input_name))
                continue;
            resize_tasks.push_back(std::async(std::launch::async, [&view, &conf, &input_name] {
                ByteImage::Ptr scaled = view.load_byte_image(conf.image_embedding);
                for (int s = 0; s < conf.input_scale; ++s)
                    scaled = rescale_half_size_gaussian(scaled);
                save_png_u8(view.directory + "/" + input_name + ".png", scaled);
            }));
        }
        for (auto& t : resize_tasks)
            t.get();
    }

    // one task per reference view, :658-733
    Clock::time_point const t0 = Clock::now();
    std::vector<std::future<void>> results;
    {
        ViewQueue queue(conf.num_devices, conf.views_in_flight);
        for (std::size_t v = 0; v < final_list.size(); ++v) {
            int const id = final_list[v];
            results.push_back(queue.add_task([&, v, id](ViewQueue::Slot const& slot) {
                SceneView const& view = views[(std::size_t)id];
                StereoView::Ptr main_view = StereoView::create(view.id,
                    view.load_byte_image(input_name), view.camera,
                    conf.use_shading, conf.gamma_correction);
                std::vector<StereoView::Ptr> stereo_views;
                for (std::size_t n = 0; n < conf.num_neighbors && n < neighbors[v].size();
                     ++n) {
                    SceneView const& nv = views[neighbors[v][n]];
                    stereo_views.push_back(StereoView::create(nv.id,
                        nv.load_byte_image(input_name), nv.camera));
                }
                int const device = conf.first_device + slot.device;
                if (conf.use_sgm) {
                    // :693-709: reuse an smvs-sgm embedding of the right size
                    int sgm_w = main_view->get_width(), sgm_h = main_view->get_height();
                    for (int s = 0; s < conf.sgm_scale; ++s) {
                        sgm_w = (sgm_w + 1) / 2;
                        sgm_h = (sgm_h + 1) / 2;
                    }
                    int whct[4] = { 0, 0, 0, 0 };
                    bool const have = mvei_header(view.image_path("smvs-sgm"), whct)
                        && whct[0] == sgm_w && whct[1] == sgm_h;
                    if (conf.force_sgm || !have) {
                        SGMStereo::Options sgm_opts;
                        sgm_opts.scale = conf.sgm_scale;
                        sgm_opts.num_steps = 128;
                        sgm_opts.min_depth = conf.sgm_min;
                        sgm_opts.max_depth = conf.sgm_max;
                        sgm_opts.device = device;
                        (void)reconstruct_sgm_depth_for_view(sgm_opts, main_view,
                            stereo_views, bundle);
                        save_mvei(view.image_path("smvs-sgm"),
                            main_view->get_embedding("smvs-sgm"));
                    } else {
                        main_view->write_image_to_view(
                            load_mvei_float(view.image_path("smvs-sgm")), "smvs-sgm");
                    }
                }
                DepthOptimizer::Options do_opts;                      // :711-720
                do_opts.regularization = 0.01 * conf.regularization;
                do_opts.num_iterations = 5;
                do_opts.min_scale = conf.output_scale;
                do_opts.use_shading = conf.use_shading;
                do_opts.output_name = report.output_name;
                do_opts.use_sgm = conf.use_sgm;
                do_opts.full_optimization = conf.full_optimization;
                do_opts.light_surf_regularization = conf.light_surf_regularization;
                do_opts.device = device;
                DepthOptimizer optimizer(main_view, stereo_views, bundle, do_opts);
                optimizer.optimize();
                // scene->save_views(), :739: the embeddings optimize() wrote
                save_mvei(view.image_path(report.output_name),
                    main_view->get_embedding(report.output_name));
                save_mvei(view.image_path(report.output_name + "N"),
                    main_view->get_embedding(report.output_name + "N"));
            }));
        }
        queue.wait_idle();
    }
    for (auto& r : results)
        r.get();
    report.reconstructed = final_list;
    report.seconds = std::chrono::duration<double>(Clock::now() - t0).count();
    return report;
}

} // namespace smvs_amd
