// C wrapper around the REFERENCE's own Boruvka implementation (compiled from where it lies under /root/reference by
// oracle/Makefile into oracle/_ref/libboruvka_ref.so -- test infrastructure, never shipped with the product).
// Mirrors the call sequence of the reference's host wrapper (lib_tree_filter/src/mst/mst.cu:41-84: createGraph, fill the
// edge array, boruvkaMST).
#include "mst/boruvka.hpp"

extern "C" void ref_boruvka(int vertex_count, int edge_count, const int* edge_index, const float* edge_weight,
                            int* edge_out) {
  struct Graph* g = createGraph(vertex_count, edge_count);
  for (int i = 0; i < edge_count; ++i) {
    g->edge[i].src = edge_index[2 * i];
    g->edge[i].dest = edge_index[2 * i + 1];
    g->edge[i].weight = edge_weight[i];
  }
  boruvkaMST(g, edge_out);
  delete[] g->edge;
  delete g;
}
