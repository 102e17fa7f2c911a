// One task per reference view on a pool of host threads bound to GPUs.
//
// The reference runs one ThreadPool task per reference view
// (app/smvsrecon.cc:558, 658-733; lib/thread_pool.h:23-159): StereoViews, the
// SGM initialisation and DepthOptimizer::optimize of a view form one task, and
// as many tasks run at once as the host has cores.  Here the pool is cut for
// GPUs: worker i drives device i % num_devices, and a device has
// `views_in_flight` workers, so that one view's host-side work (image
// conversion, grid surgery between Newton batches, uploads and the map
// download) overlaps another view's kernels -- a single view keeps the GPU busy
// for about a fifth of its wall time.  The Newton loops of two views on one
// GPU still take turns (the resident solver owns every CU); everything else
// runs concurrently on the views' own streams.
#pragma once

#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <vector>

namespace smvs_amd {

class ViewQueue
{
public:
    struct Slot
    {
        int worker;   // 0 .. num_workers() - 1
        int device;   // HIP device this worker drives (DepthOptimizer::Options::device)
    };
    typedef std::function<void(Slot const&)> Task;

    // num_devices >= 1 GPUs of this process, views_in_flight >= 1 workers each
    ViewQueue(int num_devices, int views_in_flight);
    ~ViewQueue(void);   // waits for the queued tasks
    ViewQueue(ViewQueue const&) = delete;
    ViewQueue& operator=(ViewQueue const&) = delete;

    // like ThreadPool::add_task (lib/thread_pool.h:96-118): the future carries
    // the task's exception, if any
    std::future<void> add_task(Task task);
    void wait_idle(void);
    int num_workers(void) const { return (int)workers.size(); }

private:
    void run(Slot slot);

    std::vector<std::thread> workers;
    std::deque<std::packaged_task<void(Slot const&)>> tasks;
    std::mutex lock;
    std::condition_variable wake, idle;
    int busy = 0;
    bool stopping = false;
};

} // namespace smvs_amd
