// nnc_huffman.hip -- Huffman code lengths from the index histogram (host function).
#include "nnc_common.hpp"
#include <queue>

// ======================================================================================
// 5. Huffman code lengths (host)
// ======================================================================================
extern "C" int nnc_huffman_lengths(const int64_t *counts, int32_t k, uint8_t *lengths_out, int64_t *hist_out,
                                   int64_t *total_bits_out)
{
    if (!counts || k <= 0 || !lengths_out) return fail(NNC_EINVAL, "nnc_huffman_lengths: bad argument");
    struct Node { int64_t w; int32_t minsym; int32_t left, right; };
    std::vector<Node> nodes;
    nodes.reserve(2 * (size_t)k);
    auto cmp = [&](int a, int b) {
        if (nodes[a].w != nodes[b].w) return nodes[a].w > nodes[b].w;
        return nodes[a].minsym > nodes[b].minsym;
    };
    std::priority_queue<int, std::vector<int>, decltype(cmp)> pq(cmp);
    for (int s = 0; s < k; s++) {
        lengths_out[s] = 0;
        if (counts[s] < 0) return fail(NNC_EINVAL, "nnc_huffman_lengths: negative count");
        if (counts[s] > 0) { nodes.push_back({counts[s], s, -1, -1}); pq.push((int)nodes.size() - 1); }
    }
    if (pq.size() == 1) lengths_out[nodes[pq.top()].minsym] = 1;
    else if (pq.size() > 1) {
        while (pq.size() > 1) {
            int a = pq.top(); pq.pop();
            int b = pq.top(); pq.pop();
            nodes.push_back({nodes[a].w + nodes[b].w, std::min(nodes[a].minsym, nodes[b].minsym), a, b});
            pq.push((int)nodes.size() - 1);
        }
        // depth of every leaf
        std::vector<std::pair<int, int>> stack;
        stack.push_back({pq.top(), 0});
        while (!stack.empty()) {
            auto [n, d] = stack.back();
            stack.pop_back();
            if (nodes[n].left < 0) { lengths_out[nodes[n].minsym] = (uint8_t)std::min(d, 255); continue; }
            stack.push_back({nodes[n].left, d + 1});
            stack.push_back({nodes[n].right, d + 1});
        }
    }
    if (hist_out) for (int i = 0; i < 65; i++) hist_out[i] = 0;
    int64_t total = 0;
    for (int s = 0; s < k; s++) {
        if (hist_out) hist_out[std::min<int>(lengths_out[s], 64)]++;
        total += (int64_t)lengths_out[s] * counts[s];
    }
    if (total_bits_out) *total_bits_out = total;
    return NNC_OK;
}

