// MakeBag -- one bag of instances for an image: regions from a file (-r) or sampled on the mask (-n), 8 features at every scale.
// Flags, files and exit codes of the reference's tools/MakeBag.cxx; the body the three bag tools
// share is ife/Host/BagTool.h.
#include "ife/Host/BagTool.h"

int main(int argc, char *argv[]) { return ife::host::bag_main(argc, argv, ife::host::BAG_SAMPLED, "MakeBag"); }
