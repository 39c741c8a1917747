// MakeBagDense -- one bag of instances for an image: one region per mask voxel whose box fits the image (tools/MakeBagDense.cxx:239-250), 8 features at every scale.
// Flags, files and exit codes of the reference's tools/MakeBagDense.cxx; the body the three bag tools
// share is ife/Host/BagTool.h.
#include "ife/Host/BagTool.h"

int main(int argc, char *argv[]) { return ife::host::bag_main(argc, argv, ife::host::BAG_DENSE, "MakeBagDense"); }
