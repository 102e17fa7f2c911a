#include "view_queue.h"

#include <stdexcept>

namespace smvs_amd {

ViewQueue::ViewQueue(int num_devices, int views_in_flight)
{
    if (num_devices < 1 || views_in_flight < 1)
        throw std::invalid_argument("ViewQueue: needs a device and a worker");
    // worker order: one worker per device first, then the second of each, so
    // that a short queue still spreads over the GPUs
    for (int k = 0; k < views_in_flight; ++k)
        for (int d = 0; d < num_devices; ++d) {
            Slot const slot = { (int)workers.size(), d };
            workers.emplace_back([this, slot]() { this->run(slot); });
        }
}

ViewQueue::~ViewQueue(void)
{
    {
        std::unique_lock<std::mutex> guard(lock);
        stopping = true;
    }
    wake.notify_all();
    for (std::thread& t : workers)
        t.join();
}

std::future<void>
ViewQueue::add_task(Task task)
{
    std::packaged_task<void(Slot const&)> job(std::move(task));
    std::future<void> result = job.get_future();
    {
        std::unique_lock<std::mutex> guard(lock);
        if (stopping)
            throw std::runtime_error("ViewQueue: add_task on a stopped queue");
        tasks.push_back(std::move(job));
    }
    wake.notify_one();
    return result;
}

void
ViewQueue::wait_idle(void)
{
    std::unique_lock<std::mutex> guard(lock);
    idle.wait(guard, [this]() { return tasks.empty() && busy == 0; });
}

void
ViewQueue::run(Slot slot)
{
    for (;;) {
        std::packaged_task<void(Slot const&)> job;
        {
            std::unique_lock<std::mutex> guard(lock);
            wake.wait(guard, [this]() { return stopping || !tasks.empty(); });
            if (tasks.empty())
                return;   // stopping and drained
            job = std::move(tasks.front());
            tasks.pop_front();
            busy += 1;
        }
        job(slot);   // (exceptions travel in the future)
        {
            std::unique_lock<std::mutex> guard(lock);
            busy -= 1;
            if (tasks.empty() && busy == 0)
                idle.notify_all();
        }
    }
}

} // namespace smvs_amd
