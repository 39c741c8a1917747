// MakeBagOnlyIntensity -- one bag of instances for an image: regions as MakeBag, the image intensity binned by ONE histogram (tools/MakeBagOnlyIntensity.cxx:326-389).
// Flags, files and exit codes of the reference's tools/MakeBagOnlyIntensity.cxx; the body the three bag tools
// share is ife/Host/BagTool.h.
#include "ife/Host/BagTool.h"

int main(int argc, char *argv[]) { return ife::host::bag_main(argc, argv, ife::host::BAG_ONLY_INTENSITY, "MakeBagOnlyIntensity"); }
