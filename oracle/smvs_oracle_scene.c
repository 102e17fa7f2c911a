/* TEST INFRASTRUCTURE ONLY (see smvs_oracle.h): CPU restatement of the
 * reference's neighbour-view selection, smvs::ViewSelection
 * (lib/view_selection.cc:14-161), SURVEY.md 8(f)-4.  Plain C, the reference's
 * containers emulated literally: std::map<float, size_t> as a sorted array
 * whose insert replaces an equal key's value, std::multimap<size_t, size_t,
 * std::greater> as a sorted array whose insert goes behind the equal keys.
 * Parity unpinned (the reference holds no test or vector for it); MVE
 * semantics [MVE-unverified], tests/golden/README.md M13, M15, M17, M22, M23. */
#include "smvs_oracle.h"
#include "smvs_oracle_opt.h"

#include <math.h>
#include <stdlib.h>

/* CameraInfo::fill_camera_pos: -R^T t */
static void
scene_camera_pos(const orc_scene_view *v, float *pos)
{
    for (int c = 0; c < 3; ++c)
    {
        float acc = 0.0f;
        for (int r = 0; r < 3; ++r)
            acc += v->rot[3 * r + c] * v->trans[r];
        pos[c] = -acc;
    }
}

/* (fill_world_to_cam).mult(pos, 1)[2]: inner product of the third row of R
 * with pos from 0.0f left to right, then + trans[2] * 1 */
static float
scene_depth_of(const orc_scene_view *v, const float *pos)
{
    float acc = 0.0f;
    for (int c = 0; c < 3; ++c)
        acc += v->rot[6 + c] * pos[c];
    return acc + v->trans[2] * 1.0f;
}

/* fill_inverse_calibration(...)[0] with ppoint = 0.5, paspect = 1 */
static float
scene_inv_focal(const orc_scene_view *v)
{
    float const dim = (float)(v->width > v->height ? v->width : v->height);
    return 1.0f / (v->flen * dim);
}

static int
feature_has_view(const orc_bundle *b, int f, int view_id)
{
    for (int k = b->ref_offsets[f]; k < b->ref_offsets[f + 1]; ++k)
        if (b->ref_views[k] == view_id)
            return 1;
    return 0;
}

/* view_selection.cc:134-159; returns the number of entries in out */
static int
scene_sorted_neighbors(int n_views, const orc_scene_view *views, int view,
    int *out)
{
    float main_pos[3];
    scene_camera_pos(&views[view], main_pos);
    float *keys = (float *)malloc(sizeof(float) * (size_t)(n_views + 1));
    int count = 0;
    for (int i = 0; i < n_views; ++i)
    {
        if (!views[i].present || i == view)
            continue;
        if (views[i].flen == 0.0f)
            continue;
        float pos[3];
        scene_camera_pos(&views[i], pos);
        float d[3] = { main_pos[0] - pos[0], main_pos[1] - pos[1],
            main_pos[2] - pos[2] };
        float n2 = 0.0f;
        for (int c = 0; c < 3; ++c)
            n2 += d[c] * d[c];
        float const dist = sqrtf(n2);
        /* distances[dist] = i */
        int at = 0;
        while (at < count && keys[at] < dist)
            at += 1;
        if (at < count && keys[at] == dist)
        {
            out[at] = i;
            continue;
        }
        for (int k = count; k > at; --k)
        {
            keys[k] = keys[k - 1];
            out[k] = out[k - 1];
        }
        keys[at] = dist;
        out[at] = i;
        count += 1;
    }
    free(keys);
    return count;
}

/* view_selection.cc:23-97 */
static int
scene_bundle_based(int n_views, const orc_scene_view *views,
    const orc_bundle *bundle, int view, int num_neighbors, int *out)
{
    const orc_scene_view *main_view = &views[view];
    if (!main_view->has_image)
        return 0;
    float const main_iproj0 = scene_inv_focal(main_view);

    /* list of features for the main view */
    int *mine = (int *)malloc(sizeof(int) * (size_t)(bundle->num_features + 1));
    float *footprints = (float *)malloc(sizeof(float)
        * (size_t)(bundle->num_features + 1));
    int n_mine = 0;
    for (int f = 0; f < bundle->num_features; ++f)
        if (feature_has_view(bundle, f, main_view->id))
        {
            mine[n_mine] = f;
            footprints[n_mine] = scene_depth_of(main_view,
                bundle->positions + 3 * f) * main_iproj0;
            n_mine += 1;
        }

    /* common features in the neighbouring views */
    int *neighbors = (int *)malloc(sizeof(int) * (size_t)(n_views + 1));
    int const n_sorted = scene_sorted_neighbors(n_views, views, view, neighbors);
    size_t *mm_key = (size_t *)malloc(sizeof(size_t) * (size_t)(n_views + 1));
    int *mm_val = (int *)malloc(sizeof(int) * (size_t)(n_views + 1));
    int mm_count = 0;
    for (int i = 0; i < n_sorted && i < 50; ++i)
    {
        const orc_scene_view *v = &views[neighbors[i]];
        int const id = v->id;
        if (id == view || v->flen == 0.0f || !v->has_image)
            continue;
        float const iproj0 = scene_inv_focal(v);
        size_t num_matches = 0;
        for (int f = 0; f < n_mine; ++f)
            if (feature_has_view(bundle, mine[f], v->id))
            {
                float const fp = scene_depth_of(v,
                    bundle->positions + 3 * mine[f]) * iproj0;
                float const lo = fp < footprints[f] ? fp : footprints[f];
                float const hi = fp < footprints[f] ? footprints[f] : fp;
                if (lo / hi > 0.6)
                    num_matches++;
            }
        /* multimap insert (std::greater): behind every key >= the new one */
        int at = 0;
        while (at < mm_count && mm_key[at] >= num_matches)
            at += 1;
        for (int k = mm_count; k > at; --k)
        {
            mm_key[k] = mm_key[k - 1];
            mm_val[k] = mm_val[k - 1];
        }
        mm_key[at] = num_matches;
        mm_val[at] = id;
        mm_count += 1;
    }

    /* views with the most common features */
    int n_out = 0;
    for (int k = 0; k < mm_count; ++k)
    {
        if (mm_key[k] > 10)
            out[n_out++] = mm_val[k];
        if (n_out >= num_neighbors)
            break;
    }
    free(mine); free(footprints); free(neighbors); free(mm_key); free(mm_val);
    return n_out;
}

/* view_selection.cc:99-132 */
static int
scene_position_based(int n_views, const orc_scene_view *views, int view,
    int *out)
{
    const orc_scene_view *m = &views[view];
    /* fill_viewing_direction [MVE-unverified]: third row of rot */
    float const main_dir[3] = { m->rot[6], m->rot[7], m->rot[8] };
    float const main_up[3] = { m->rot[2], m->rot[5], m->rot[8] };
    int *neighbors = (int *)malloc(sizeof(int) * (size_t)(n_views + 1));
    int const n_sorted = scene_sorted_neighbors(n_views, views, view, neighbors);
    int n_out = 0;
    for (int k = 0; k < n_sorted; ++k)
    {
        const orc_scene_view *c = &views[neighbors[k]];
        float const dir[3] = { c->rot[6], c->rot[7], c->rot[8] };
        float const up[3] = { c->rot[2], c->rot[5], c->rot[8] };
        float up_dot = 0.0f, dir_dot = 0.0f;
        for (int a = 0; a < 3; ++a)
        {
            up_dot += main_up[a] * up[a];
            dir_dot += main_dir[a] * dir[a];
        }
        if (up_dot < 0 || dir_dot < 0.65)
            continue;   /* erased */
        out[n_out++] = neighbors[k];
    }
    free(neighbors);
    return n_out;
}

/* ViewSelection::get_neighbors_for_view, view_selection.cc:14-21.  out holds
 * up to n_views indices into the view list; returns how many. */
int
orc_select_neighbors(int n_views, const orc_scene_view *views,
    const orc_bundle *bundle, int view, int num_neighbors, int *out)
{
    if (bundle != NULL)
        return scene_bundle_based(n_views, views, bundle, view, num_neighbors,
            out);
    return scene_position_based(n_views, views, view, out);
}
