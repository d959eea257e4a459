// TEST INFRASTRUCTURE (oracle/): golden-vector generator for the pair list.
// Our own driver; it includes the reference's HEADERS from /root/reference/src (never copied) and is
// linked with the reference objects built by oracle/Makefile.  No C-ABI call of the reference exposes
// edge indices (engine_c_library.h:12-32), so this instantiates
// InteractionGraph<preferred_bead_type> (interaction_graph.h:261, bead_interaction.h:221) on
// /input/potential/rotamer/pair_interaction and dumps edge_indices1/2, edge_id1/2, edge_value.
// The pair functors of the asymmetric graphs (hbond.cpp:152-286, environment.cpp:12-69) live in anonymous namespaces of
// their translation units, so those two reference files are INCLUDED here from /root/reference/src (and their objects left
// out of the link, oracle/Makefile): the node classes then are nameable, and the edge lists of the engine's own
// protein_hbond / hbond_coverage* / environment_coverage nodes are dumped after the force pass, one section per node:
//   graph <node name> n_edge <n>      followed by n lines  "i1 i2"
//
// usage: pairlist_dump config.up out.txt
#include "deriv_engine.h"
#include "interaction_graph.h"
#include "bead_interaction.h"
#include "hbond.cpp"
#include "environment.cpp"
#include <cstdio>
using namespace h5;
using namespace std;

int main(int argc, char** argv) try {
    if(argc<3) {fprintf(stderr,"usage: %s config.up out.txt\n", argv[0]); return 2;}
    H5Obj config = h5_obj(H5Fclose, H5Fopen(argv[1], H5F_ACC_RDONLY, H5P_DEFAULT));
    auto pos_shape = get_dset_size(3, config.get(), "/input/pos");
    int n_atom = pos_shape[0];
    auto potential_group = open_group(config.get(), "/input/potential");
    DerivEngine engine = initialize_engine_from_hdf5(n_atom, potential_group.get(), true);
    traverse_dset<3,float>(config.get(), "/input/pos", [&](size_t na, size_t d, size_t ns, float x) {
            engine.pos->output(d,na) = x;});
    engine.compute(PotentialAndDerivMode);

    auto rot_grp = open_group(potential_group.get(), "rotamer");
    auto args = read_attribute<vector<string>>(potential_group.get(), "rotamer", "arguments");
    auto& bead_node = engine.get_computation<CoordNode>(args[0]);
    auto pg = open_group(rot_grp.get(), "pair_interaction");
    InteractionGraph<preferred_bead_type> ig(pg.get(), &bead_node);
    FILE* f = fopen(argv[2], "w");
    for(int pass=0; pass<2; ++pass) {   // second call goes through the cached (non-rebuild) path
        ig.compute_edges();
        fprintf(f, "n_edge %i cutoff %.9g pass %i\n", ig.n_edge, ig.cutoff, pass);
        for(int ne=0; ne<ig.n_edge; ++ne)
            fprintf(f, "%i %i %i %i %.9g\n", ig.edge_indices1[ne], ig.edge_indices2[ne],
                    ig.edge_id1[ne], ig.edge_id2[ne], ig.edge_value[ne]);
    }
    // the asymmetric graphs, straight from the engine's nodes (state of the force pass above)
    for(auto& n : engine.nodes) {
        auto* c = n.computation.get();
        int n_edge = -1; const int *i1 = nullptr, *i2 = nullptr;
        if(auto* p = dynamic_cast<ProteinHBond*>(c))             {n_edge = p->igraph.n_edge; i1 = p->igraph.edge_indices1; i2 = p->igraph.edge_indices2;}
        else if(auto* p = dynamic_cast<HBondCoverage*>(c))       {n_edge = p->igraph.n_edge; i1 = p->igraph.edge_indices1; i2 = p->igraph.edge_indices2;}
        else if(auto* p = dynamic_cast<EnvironmentCoverage*>(c)) {n_edge = p->igraph.n_edge; i1 = p->igraph.edge_indices1; i2 = p->igraph.edge_indices2;}
        if(n_edge < 0) continue;
        fprintf(f, "graph %s n_edge %i\n", n.name.c_str(), n_edge);
        for(int ne=0; ne<n_edge; ++ne) fprintf(f, "%i %i\n", i1[ne], i2[ne]);
    }
    fclose(f);
    return 0;
} catch(const string& e) {
    fprintf(stderr, "ERROR: %s\n", e.c_str());
    return 1;
}
