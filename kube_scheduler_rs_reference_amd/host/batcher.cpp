#include "batcher.hpp"

#include <algorithm>
#include <stdexcept>

namespace ksched_host {

PodBatcher::PodBatcher(size_t max_pods) : max_pods_(std::max<size_t>(max_pods, 1)) {}

bool PodBatcher::push(PodPtr pod) {
    if (!pod) return false;
    const std::string key = full_name(pod->metadata);
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (closed_) return false;
        auto it = ticket_of_.find(key);
        if (it != ticket_of_.end()) {
            // still queued: the newer object takes the older one's place in line (tickets are consecutive from the queue's head)
            queue_[(size_t)(it->second - head_ticket_)].second = std::move(pod);
            ++coalesced_;
            return true;
        }
        ticket_of_.emplace(key, next_ticket_);
        queue_.emplace_back(next_ticket_++, std::move(pod));
    }
    cv_.notify_one();
    return true;
}

void PodBatcher::close() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
    }
    cv_.notify_all();
}

bool PodBatcher::closed() const {
    std::lock_guard<std::mutex> lk(mu_);
    return closed_;
}

size_t PodBatcher::pending() const {
    std::lock_guard<std::mutex> lk(mu_);
    return queue_.size();
}

uint64_t PodBatcher::coalesced() const {
    std::lock_guard<std::mutex> lk(mu_);
    return coalesced_;
}

std::vector<PodBatcher::PodPtr> PodBatcher::take_locked() {
    std::vector<PodPtr> out;
    const size_t k = std::min(max_pods_, queue_.size());
    out.reserve(k);
    for (size_t i = 0; i < k; ++i) {
        ticket_of_.erase(full_name(queue_.front().second->metadata));
        out.push_back(std::move(queue_.front().second));
        queue_.pop_front();
        ++head_ticket_;
    }
    return out;
}

std::vector<PodBatcher::PodPtr> PodBatcher::next_batch() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !queue_.empty() || closed_; });
    std::vector<PodPtr> out = take_locked();
    if (!queue_.empty()) {  // more than one batch's worth is ready: another consumer need not wait for the next push
        lk.unlock();
        cv_.notify_one();
    }
    return out;
}

std::vector<PodBatcher::PodPtr> PodBatcher::try_next_batch() {
    std::lock_guard<std::mutex> lk(mu_);
    return take_locked();
}

BatchLoopStats run_batches(PodBatcher &batcher,
                           const std::function<std::vector<ReconcileOutcome>(const std::vector<const corev1::Pod *> &)> &reconcile,
                           const std::function<void(const PodBatcher::PodPtr &, const ReconcileOutcome &)> &done,
                           const std::function<void(const PodBatcher::PodPtr &, const std::string &)> &failed) {
    BatchLoopStats st;
    for (;;) {
        const std::vector<PodBatcher::PodPtr> batch = batcher.next_batch();
        if (batch.empty()) return st;  // closed and drained
        std::vector<const corev1::Pod *> raw;
        raw.reserve(batch.size());
        for (const auto &p : batch) raw.push_back(p.get());
        std::vector<ReconcileOutcome> out;
        bool whole = true;
        try {
            out = reconcile(raw);
        } catch (const PodEncodeError &) {
            // a pod of the batch cannot be encoded: thrown by Snapshot::encode_pods, i.e. before anything of the batch was evaluated
            // or POSTed -- the ONE failure after which the batch's pods may be reconciled again, one at a time.  Every other
            // exception propagates: once bindings may have been created, re-reconciling the batch would POST them a second time.
            if (!failed) throw;
            whole = false;
        }
        if (!whole) {  // isolate the offender(s): the batch's pods one at a time
            ++st.isolated_batches;
            for (const auto &p : batch) {
                try {
                    const std::vector<ReconcileOutcome> one = reconcile({p.get()});
                    if (one.size() != 1) throw std::logic_error("run_batches: the reconcile function must return one outcome per pod");
                    done(p, one[0]);
                } catch (const PodEncodeError &e) {
                    failed(p, e.what());
                    ++st.failed_pods;
                }
            }
            ++st.batches;
            st.pods += batch.size();
            st.largest = std::max<uint64_t>(st.largest, batch.size());
            continue;
        }
        if (out.size() != batch.size()) throw std::logic_error("run_batches: the reconcile function must return one outcome per pod");
        for (size_t i = 0; i < batch.size(); ++i) done(batch[i], out[i]);
        ++st.batches;
        st.pods += batch.size();
        st.largest = std::max<uint64_t>(st.largest, batch.size());
    }
}

}  // namespace ksched_host
