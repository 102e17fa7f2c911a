// Installs the host mirror's image-memory hook (host/image.h): images of a
// megabyte and more live in page-locked memory from the device library's pool
// (smvs_pinned_alloc, include/smvs_hip.h), so that the views' u8 images go to the
// device and the depth / normal maps come back as one DMA each.  Without a
// device (the CPU-only tests of the host logic) the first allocation fails and
// the hook retires: images are plain vectors again.
#include <atomic>
#include <cstdlib>

#include "image.h"
#include "../../../include/smvs_hip.h"

namespace {

std::atomic<bool> g_retired{ false };

void*
pinned_alloc(std::size_t bytes)
{
    if (g_retired.load(std::memory_order_relaxed))
        return nullptr;
    void* p = nullptr;
    if (smvs_pinned_alloc(bytes, &p) != SMVS_OK) {
        // no device / no pinned memory left: stop asking
        g_retired.store(true, std::memory_order_relaxed);
        return nullptr;
    }
    return p;
}

void
pinned_free(void* p)
{
    (void)smvs_pinned_free(p);
}

struct Install
{
    Install(void)
    {
        // SMVS_PAGEABLE_IMAGES=1: A/B switch (the staging-copy path)
        if (std::getenv("SMVS_PAGEABLE_IMAGES") != nullptr)
            return;
        smvs_amd::ImageMemory::alloc_hook() = pinned_alloc;
        smvs_amd::ImageMemory::free_hook() = pinned_free;
    }
} g_install;

} // namespace
