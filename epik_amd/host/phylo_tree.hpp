// phylo_tree.hpp -- Newick in, post-order ids, jplace Newick out.
//
// Mirrors the i2l surface EPIK uses (absent submodule; contract from the call sites):
//   i2l::io::parse_newick(string)                          main.cpp:294
//   i2l::io::to_newick(tree, /*jplace=*/true)              main.cpp:296-297
//   phylo_tree::get_node_count(), get_by_postorder_id(id)  place.cpp:92-103, 429
//   phylo_node::get_branch_length()                        place.cpp:110, 435
//   db.tree_index()[id].{subtree_num_nodes, subtree_total_length}   place.cpp:113-114
// ASSUMPTIONS: post-order ids 0..N-1 in the order of the Newick string, root = N-1;
// jplace output is `label:length{postorder_id}`.
#ifndef EPIK_AMD_HOST_PHYLO_TREE_HPP
#define EPIK_AMD_HOST_PHYLO_TREE_HPP

#include <cstddef>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

namespace epik_amd {

class phylo_node {
public:
    using id_type = unsigned int;
    using branch_length_type = double;
    branch_length_type get_branch_length() const noexcept { return branch_length; }
    const std::string& get_label() const noexcept { return label; }

    std::string label;
    branch_length_type branch_length = 0.0;
    int parent = -1;
    std::vector<int> children;
    id_type postorder_id = 0;
};

struct tree_index_entry {
    size_t subtree_num_nodes = 0;       // nodes in the subtree, the node included
    double subtree_total_length = 0.0;  // branch lengths strictly below the node
};

class phylo_tree {
public:
    size_t get_node_count() const noexcept { return _nodes.size(); }
    std::optional<const phylo_node*> get_by_postorder_id(phylo_node::id_type id) const noexcept
    {
        if (id >= _nodes.size()) return std::nullopt;
        return &_nodes[id];
    }
    const std::vector<phylo_node>& nodes() const noexcept { return _nodes; }
    std::vector<tree_index_entry> tree_index() const;

    std::vector<phylo_node> _nodes;  // indexed by post-order id
};

namespace io {
phylo_tree parse_newick(std::string_view newick);
std::string to_newick(const phylo_tree& tree, bool jplace = false);
}  // namespace io

}  // namespace epik_amd
#endif
