#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "phylo_tree.hpp"

namespace epik_amd {

std::vector<tree_index_entry> phylo_tree::tree_index() const
{
    std::vector<tree_index_entry> index(_nodes.size());
    for (size_t i = 0; i < _nodes.size(); ++i) {  // post-order: children come first
        index[i].subtree_num_nodes += 1;
        for (int c : _nodes[i].children) {
            index[i].subtree_num_nodes += index[(size_t)c].subtree_num_nodes;
            index[i].subtree_total_length += index[(size_t)c].subtree_total_length + _nodes[(size_t)c].branch_length;
        }
    }
    return index;
}

namespace io {

namespace {

struct parser {
    std::string_view s;
    size_t pos = 0;
    phylo_tree tree;

    void skip_ws()
    {
        while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\n' || s[pos] == '\t' || s[pos] == '\r')) ++pos;
    }
    [[noreturn]] void fail(const char* what) const
    {
        throw std::runtime_error(std::string("Newick parse error at ") + std::to_string(pos) + ": " + what);
    }
    // parses one subtree, appends its nodes in post-order, returns the id of its root
    int subtree()
    {
        skip_ws();
        std::vector<int> children;
        if (pos < s.size() && s[pos] == '(') {
            ++pos;
            while (true) {
                children.push_back(subtree());
                skip_ws();
                if (pos >= s.size()) fail("unbalanced parenthesis");
                if (s[pos] == ',') {
                    ++pos;
                    continue;
                }
                if (s[pos] == ')') {
                    ++pos;
                    break;
                }
                fail("expected ',' or ')'");
            }
        }
        phylo_node node;
        skip_ws();
        if (pos < s.size() && (s[pos] == '\'' || s[pos] == '"')) {
            const char q = s[pos++];
            while (pos < s.size() && s[pos] != q) node.label.push_back(s[pos++]);
            if (pos >= s.size()) fail("unterminated quoted label");
            ++pos;
        } else {
            while (pos < s.size() && s[pos] != ':' && s[pos] != ',' && s[pos] != ')' && s[pos] != '(' &&
                   s[pos] != ';' && s[pos] != '{' && s[pos] != '[')
                node.label.push_back(s[pos++]);
            while (!node.label.empty() && (node.label.back() == ' ' || node.label.back() == '\n')) node.label.pop_back();
        }
        skip_ws();
        if (pos < s.size() && s[pos] == ':') {
            ++pos;
            skip_ws();
            const std::string num(s.substr(pos, 64));
            char* end = nullptr;
            node.branch_length = std::strtod(num.c_str(), &end);
            if (end == num.c_str()) fail("expected a branch length");
            pos += (size_t)(end - num.c_str());
        }
        skip_ws();
        while (pos < s.size() && (s[pos] == '{' || s[pos] == '[')) {  // jplace edge ids / comments
            const char close = s[pos] == '{' ? '}' : ']';
            while (pos < s.size() && s[pos] != close) ++pos;
            if (pos < s.size()) ++pos;
            skip_ws();
        }
        const int id = (int)tree._nodes.size();
        node.postorder_id = (phylo_node::id_type)id;
        node.children = children;
        tree._nodes.push_back(std::move(node));
        for (int c : children) tree._nodes[(size_t)c].parent = id;
        return id;
    }
};

void emit(const phylo_tree& tree, int id, bool jplace, std::string& out)
{
    const phylo_node& n = tree._nodes[(size_t)id];
    if (!n.children.empty()) {
        out.push_back('(');
        for (size_t i = 0; i < n.children.size(); ++i) {
            if (i) out.push_back(',');
            emit(tree, n.children[i], jplace, out);
        }
        out.push_back(')');
    }
    out += n.label;
    char buf[64];
    std::snprintf(buf, sizeof buf, ":%.10g", n.branch_length);
    out += buf;
    if (jplace) {
        std::snprintf(buf, sizeof buf, "{%u}", n.postorder_id);
        out += buf;
    }
}

}  // namespace

phylo_tree parse_newick(std::string_view newick)
{
    parser p;
    p.s = newick;
    p.subtree();
    p.skip_ws();
    if (p.pos < newick.size() && newick[p.pos] == ';') ++p.pos;
    p.skip_ws();
    if (p.pos != newick.size()) p.fail("trailing characters");
    if (p.tree._nodes.empty()) p.fail("empty tree");
    return std::move(p.tree);
}

std::string to_newick(const phylo_tree& tree, bool jplace)
{
    std::string out;
    if (!tree._nodes.empty()) emit(tree, (int)tree._nodes.size() - 1, jplace, out);
    out.push_back(';');
    return out;
}

}  // namespace io
}  // namespace epik_amd
